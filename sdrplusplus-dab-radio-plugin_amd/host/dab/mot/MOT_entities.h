// dab/mot/MOT_entities.h -- the one name render_radio_block.h needs from the reference's MOT layer
// (/root/reference/src/render_radio_block.h:8, :28).  Multimedia object transfer (slideshows) is not part of the
// hot path and is not provided.
#pragma once
#include <cstdint>
typedef uint16_t mot_transport_id_t;
