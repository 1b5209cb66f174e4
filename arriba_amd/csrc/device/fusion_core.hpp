// arriba_amd/csrc/device/fusion_core.hpp -- candidate building (find_fusions) on the device; filled in below.
#ifndef AGPU_FUSION_CORE_HPP
#define AGPU_FUSION_CORE_HPP 1
#include "filter_core.hpp"
namespace agpu {
struct FusionEmission { uint32_t placeholder; };
}
#endif
