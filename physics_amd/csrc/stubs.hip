// stubs.hip — placeholders for translation units not written yet (removed as each one lands)
#include "kernels.hpp"
namespace phys {
int32_t halo_pack(phys_world*, float, float, void*, uint64_t, uint64_t*) { return PHYS_ERR_UNSUPPORTED; }
int32_t halo_pairs(phys_world*, const void*, uint64_t, uint64_t*) { return PHYS_ERR_UNSUPPORTED; }
}
