// xq_internal.hpp — declarations shared between the translation units of libxq_ops.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "xq_common.hpp"

extern thread_local char g_err[512];
int xq_set_error(int code, const char *fmt, const char *a = "", long b = 0, long c = 0);
int xq_check_launch(const char *what);
int num_cus();
// true the first time it is called with `mask` on the CURRENT device (<= 64 devices per process): guards the once-per-device calls of
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) — a process-wide `static bool` set the attribute on the first device only (round-4 advisor)
bool first_call_on_this_device(unsigned long long *mask);
int check_common(const char *fn, const void *z, int B, int C, int HW, const void *E, int V);

static inline int chunk_codes(int C) { return C == 8 ? 256 : (C <= 64 ? 128 : 64); }

struct AssignWs {
    float *wb;                 // [Vpad/32][C/8][64 lanes][4]  fragment-ordered ehat
    float *ee;                 // [Vpad]
    unsigned long long *keys;  // [N]
    float *partials;           // [4096]
    int Vpad;
};
static constexpr int MAX_PARTIALS = 4096;

static inline size_t assign_ws_layout(int64_t N, int C, int V, char *base, AssignWs *ws) {
    const int CH = chunk_codes(C);
    const int Vpad = (V + CH - 1) / CH * CH;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = xq::align_up(off + bytes, 256); return o; };
    size_t o_wb = take((size_t)Vpad * C * 4);
    size_t o_ee = take((size_t)Vpad * 4);
    size_t o_keys = take((size_t)N * 8);
    size_t o_part = take((size_t)MAX_PARTIALS * 4);
    if (ws) {
        ws->wb = (float *)(base + o_wb);
        ws->ee = (float *)(base + o_ee);
        ws->keys = (unsigned long long *)(base + o_keys);
        ws->partials = (float *)(base + o_part);
        ws->Vpad = Vpad;
    }
    return off;
}


enum { XQI_PREP = 1, XQI_SEARCH = 2 };
// prep: E -> fragment-ordered (normalised) codebook + |e|^2 in ws; search: keys[n] = min over codes of (ord(d)<<32 | code)
int launch_assign(int mode, int C, const float *z, long N, int HW, const float *E, int V, const AssignWs &ws, hipStream_t s,
                  int what = XQI_PREP | XQI_SEARCH);

// measurement hooks (xq_vq.hip): start/stop HIP events around an instrumented launch when xq_prof_enable(1) is armed;
// `work` = algorithmic flops (or bytes) of the launch, `kind` one of XQ_PROF_* (include/xq_ops.h)
namespace xq {
int prof_begin(int kind, double work, hipStream_t s);
void prof_end(int slot, hipStream_t s);
}
