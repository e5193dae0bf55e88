// basic_radio/basic_data_packet_channel.h -- what BasicRadio::Get_Data_Packet_Channel returns
// (/root/reference/src/render_radio_block.cpp:533, 593).  Packet-mode sub-channels (TMId 3) are not followed by this
// receiver (data services are outside the hot path), so none is ever created; the type exists
// for the callers that ask.
#pragma once
#include "basic_radio/basic_slideshow.h"
#include "dab/database/dab_database_entities.h"

class Basic_Data_Packet_Channel {
public:
    explicit Basic_Data_Packet_Channel(const Subchannel &subchannel) : m_subchannel(subchannel) {}
    Basic_Slideshow_Manager &GetSlideshowManager() { return m_slideshows; }
    const Subchannel &GetSubchannel() const { return m_subchannel; }

private:
    Subchannel m_subchannel;
    Basic_Slideshow_Manager m_slideshows;
};
