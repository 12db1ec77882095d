// Differential Evolution: the multi-GPU peer-exchange instantiations of the generation kernel (XM = 2, sx_de_kernel.hpp),
// a translation unit of their own so that they compile next to sx_de.hip's instead of after them.
#define SX_DE_XM 2
#include "sx_de_kernel.hpp"

namespace sx {
void *de_p2p_kernel(int fun_id, int n, int64_t P, int strategy, int constraints) {
    return (void *)pick_kernel<SX_RNG_PHILOX, 2>(fun_id, n, P, strategy, constraints);
}
}  // namespace sx
