// Weight packing for the row-block chains (tg_chain.hip): tg_pack_weights splits an (N x K) fp32 weight into bf16 hi / lo and stores it
// in MFMA fragment order (layout and device body: tg_pack.h), once per optimizer step.  The layer's prelude launch (tg_layer.hip) runs the
// same body beside its other roles; this is the stand-alone entry point.
//
// replaces: nothing of the reference by itself -- it prepares the operand of the chain kernels that replace aten::mm / addmm behind the
//           nn.Linear layers of models/modules.py:54-69, :152-163, :199-235.
#include <algorithm>

#include "tg_common.h"
#include "tg_pack.h"

namespace {
__global__ void __launch_bounds__(256) pack_weights_kernel(tgs::PackJobs jobs) { tgs::pack_body(jobs, (int)blockIdx.x, (int)gridDim.x); }
}  // namespace

namespace tg {

int64_t packed_floats(int N, int K) { return (int64_t)((N + 15) / 16) * ((K + 31) / 32) * 512; }

int pack_weights(int njobs, const tg_pack_job* jobs, hipStream_t s) {
    TG_REQUIRE(njobs >= 0 && njobs <= 24, "tg_pack_weights: at most 24 jobs per launch");
    if (njobs == 0) return TG_OK;
    tgs::PackJobs pj;
    const int total = tgs::pack_jobs_fill(pj, njobs, jobs);
    TG_REQUIRE(total >= 0, "tg_pack_weights: bad job");
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 3) / 4, 2048);
    pack_weights_kernel<<<blocks, 256, 0, s>>>(pj);
    return launch_status("pack_weights_kernel");
}

}  // namespace tg

extern "C" int64_t tg_packed_floats(int N, int K) { return tg::packed_floats(N, K); }

extern "C" int tg_pack_weights(int njobs, const tg_pack_job* jobs, void* stream) { return tg::pack_weights(njobs, jobs, (hipStream_t)stream); }
