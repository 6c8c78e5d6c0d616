// rq_tail.hip -- everything after the corpus scan in ONE launch (k <= 320).
//
// grid (chunks of 512*NV bins, B queries), 256 threads.  Each workgroup
//   A. derives the threshold T = P - 2.25 eps, P = the k-th largest partition maximum (partitions = groups of the
//      scan's per-workgroup maxima; distinct scan workgroups own distinct bins, so at least k rows reach P and the
//      k-th exact score is >= P - eps) with a 20-bit ballot radix select in wave 0,
//   B. finds the bins of its chunk whose largest score reaches T and turns them into ROW JOBS from the bin's 8-byte
//      scan record (rq_device.h): the arg-max row alone when the bound on the second-largest score is below T -- the
//      usual case --, the best two rows when only the bound on the third-largest is, all 64 rows of the bin otherwise,
//   C. re-scores the job rows exactly in fp64 (16 lanes per row, 8 rows of loads in flight per wave) and appends the
//      (score, row) keys to the query's compact candidate list with 8-byte write-through (sc1) stores,
//   D. publishes: every wave drains vmcnt, workgroup barrier, ONE lane draws a ticket (agent-scope atomic add).  The
//      workgroup that draws the last ticket of its query runs ONE agent-scope acquire (buffer_inv sc1 + vmcnt(0), then a
//      workgroup barrier), reads the keys (sc1 loads), ranks them, writes the exact top-k and the certificate, and
//      resets the counters.  No RELEASE fences: a release fence per workgroup serialises on the L2 write-back
//      (measured +60 us per launch); the producer side is write-through stores + drain + barrier + ticket.
// Exactness: a row that is not re-scored has approximate score < T (its bin's largest, second- or third-largest bound
// is below T), so its exact score is < T + eps < s_k; see rq_final_body.h and DESIGN.md 4.2.
// Replaces reference rag_uq/streaming_index.py:355-368 (collection.query + `1 - distance`) after the scan.
#include "rq_device.h"
#include "rq_kernels.h"
#include "rq_tail_body.h"

template <int NV>
__global__ __launch_bounds__(256) void rq_tail_kernel(RqTailArgs a) {
    __shared__ RqTailLds lds;
    rq_tail_body<NV>(a, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, lds);
}

template <int NV>
static hipError_t rq_tail_launch_nv(const RqTailArgs& a, int B, hipStream_t stream) {
    const int64_t chunks = (a.nbins + 512 * NV - 1) / (512 * NV);
    if (chunks < 1 || chunks > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL((rq_tail_kernel<NV>), dim3((unsigned)chunks, B), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t rq_tail_launch(const RqTailArgs& a, int B, hipStream_t stream) {
    if (a.m < 1 || a.m > RQ_FAST_MAX_M || a.k < 1 || a.k > RQ_FAST_MAX_K) return hipErrorInvalidValue;
    return rq_tail_small_chunks(a.nbins, B) ? rq_tail_launch_nv<1>(a, B, stream) : rq_tail_launch_nv<4>(a, B, stream);
}
