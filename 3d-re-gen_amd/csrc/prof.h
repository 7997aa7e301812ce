// prof.h -- optional per-kernel-family timing with HIP events on the launch stream (used by bench.py's roofline leg).
#ifndef R3G_PROF_H
#define R3G_PROF_H
#include <hip/hip_runtime.h>

namespace r3g {

enum ProfCat { PC_GEMM = 0, PC_ATTN, PC_LAYERNORM, PC_QKV_SPLIT, PC_GEMV, PC_ELEMWISE, PC_MC_CLASSIFY, PC_MC_OTHER, PC_MESH, PC_COUNT };

// RAII bracket around one kernel launch.  `work` = algorithmic FLOPs (MFMA kernels) or bytes (HBM kernels).
struct ProfScope {
    int slot;
    hipStream_t s;
    // bytes: algorithmic operand + result bytes of an MFMA launch (families whose `work` is FLOPs)
    ProfScope(int cat, double work, hipStream_t stream, double bytes = 0.0);
    ~ProfScope();
};

void prof_enable(bool on);   // enabling resets the accumulators
bool prof_enabled();
// drains outstanding events; fills per-category launch counts, total milliseconds and total work
void prof_read(long long* counts, double* ms, double* work);
void prof_read_bytes(double* bytes);   // per category: sum of the `bytes` given to ProfScope

}  // namespace r3g
#endif
