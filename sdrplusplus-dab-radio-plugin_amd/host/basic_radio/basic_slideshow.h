// basic_radio/basic_slideshow.h -- the slideshow entities the GUI asks an audio or data channel for
// (/root/reference/src/render_radio_block.cpp:309-340, 591-593).  MOT slideshows ride in the programme-associated data of
// the audio frames or in packet-mode sub-channels, both behind the audio / packet decoders that are outside the
// hot path: the manager exists and is always empty.
#pragma once
#include <cstdint>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "dab/mot/MOT_entities.h"
#include "utility/observable.h"
#include "utility/span.h"

struct Basic_Slideshow {
    mot_transport_id_t transport_id = 0;
    std::string name;
    uint32_t trigger_time = 0;
    uint32_t expire_time = 0;
    uint8_t category_id = 0;
    uint8_t slide_id = 0;
    std::string category_title;
    std::string click_through_url;
    std::string alt_location_url;
    std::vector<uint8_t> image_data;
};

class Basic_Slideshow_Manager {
public:
    std::mutex &GetSlideshowsMutex() { return m_mutex; }
    std::list<std::shared_ptr<Basic_Slideshow>> &GetSlideshows() { return m_slideshows; }
    Observable<std::shared_ptr<Basic_Slideshow> &> &OnNewSlideshow() { return m_obs_new; }
    Observable<std::shared_ptr<Basic_Slideshow> &> &OnRemoveSlideshow() { return m_obs_remove; }
    size_t GetMaxSize() const { return m_max_size; }
    void SetMaxSize(size_t n) { m_max_size = n; }

private:
    std::mutex m_mutex;
    std::list<std::shared_ptr<Basic_Slideshow>> m_slideshows;
    Observable<std::shared_ptr<Basic_Slideshow> &> m_obs_new, m_obs_remove;
    size_t m_max_size = 25;
};
