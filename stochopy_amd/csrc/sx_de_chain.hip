// Differential Evolution: the single-GPU chained-finalize instantiations of the generation kernel (XM = 1, sx_de_kernel.hpp),
// a translation unit of their own so that they compile next to sx_de.hip's instead of after them.
#define SX_DE_XM 1
#include "sx_de_kernel.hpp"

namespace sx {
void *de_chain_kernel(int fun_id, int n, int64_t P, int strategy, int constraints) {
    return (void *)pick_kernel<SX_RNG_PHILOX, 1>(fun_id, n, P, strategy, constraints);
}
}  // namespace sx
