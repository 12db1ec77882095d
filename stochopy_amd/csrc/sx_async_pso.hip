// updating="immediate": the PSO / CPSO sweeps of sx_async.hip (pso_async_kernel, sx_pso_async_generation) as a translation unit of
// their own, so that they compile next to the DE sweeps instead of after them.
#define SX_ASYNC_PART 1
#include "sx_async.hip"
