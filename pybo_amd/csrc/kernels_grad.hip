// kernels_grad.hip -- posterior moments WITH input gradients for small candidate batches
// (the `f(x[None], grad=True)` calls of the L-BFGS refinement, pybo/solvers/lbfgs.py:56-58).
#include "gpx_internal.h"
namespace gpx {
int predict_grad_host(gpx_handle* h, const double*, int64_t, double*, double*, double*, double*) {
    h->err = "predict with gradients: not built yet";
    return GPX_EARG;
}
}  // namespace gpx
