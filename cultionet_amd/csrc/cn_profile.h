// Optional per-launch timing of the contraction kernels with HIP events on the launch stream
// (diagnostics for bench.py's roofline figure; off by default, zero cost when off).
#pragma once
#include <hip/hip_runtime.h>

#define CN_PROF_KINDS 8  // 0 igemm<NT=128>  1 igemm<NT<=64>  2 wgrad<T=9>  3 wgrad<T=1>  4 bf16 conv  5 bf16 wgrad

bool cn_prof_on();
void cn_prof_before(hipStream_t stream);
void cn_prof_after(hipStream_t stream, int kind, double flops);
void cn_prof_desc(const char* fmt, ...);
void cn_prof_name(const char* fmt, ...);  // rocprof-style kernel name of the next recorded launch
void cn_prof_bytes(double bytes);      // algorithmic HBM bytes of the next recorded launch
