// kernels_cascade.hip -- the posterior domain heuristics of the device-driven cascade, gfx950 only.
// Replaces, inside the hmmsearch process launched at checkm/hmmer.py:70, the region scan of HMMER's domain definition
// ("p7_domaindef_ByPosteriorHeuristics": rt1 0.25 / rt2 0.10 / rt3 0.20 on the begin/end/occupancy posteriors) and the hand-over
// to envelope rescoring, which the host used to do between two device phases.  One wavefront per pair that passed the Forward
// filter: it turns the three decoding terms per residue the Backward parser left in the workspace into the running sums
// btot/etot (in place: the float prefix sums are part of the decision, so they are formed once, in residue order, exactly as the
// CPU restatement forms them), finds the regions, and for every region allocates the workspace of what comes next --
//   one envelope (FbWork, full = 1)             -> envelope queue of the model's register class
//   a multi-domain region (FbWork full = 2 + EnsWork) -> region queue: multihit Forward with M, I, D rows, then the trace ensemble
// and leaves a RegionRec for the host in pinned memory.  Float compares only: no transcendental.
#include <hip/hip_runtime.h>
#include "dev_types.h"
#include "cascade_dev.h"
#include "cascade_regions.h"

namespace ckm {



// list/count: a queue of parser items of `fwork` whose decoding terms are complete (the host-visible form of the stage; the cascade
// itself runs the scan inside the fused parser kernel)
__global__ void __launch_bounds__(64) region_kernel(const uint32_t *__restrict__ list, const uint32_t *__restrict__ count, uint32_t cap,
                                                   const FbWork *__restrict__ fwork, CascadeDev cd, const DevModel *__restrict__ models, float *__restrict__ ws) {
  const uint32_t n = min(*count, cap);
  const int lane = threadIdx.x;
  for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
    const FbWork w = fwork[list[k]];
    region_scan<false>(cd, models[w.model], w, ws, lane);
  }
}

void launch_regions(hipStream_t stream, uint32_t nblocks, const uint32_t *list, const uint32_t *count, uint32_t cap, const FbWork *fwork,
                    const CascadeDev &cd, const DevModel *models, float *ws) {
  if (nblocks) hipLaunchKernelGGL(region_kernel, dim3(nblocks), dim3(64), 0, stream, list, count, cap, fwork, cd, models, ws);
}

}  // namespace ckm
