#include "trace.h"
#include "../../include/mcvc.h"
#include <vector>
#include <string.h>

bool g_mcvc_trace_on = false;

namespace {
struct Rec { int kind; hipEvent_t a, b; double flops, bytes; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event()
{
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
const char* kNames[K_COUNT] = {
    "conv_direct<2,2,2,2>", "conv_direct<2,1,2,2>", "conv_direct<1,2,2,2>", "conv_direct<2,1,4,1>", "conv_direct<1,2,1,4>", "conv_direct<1,1,1,4>", "conv_direct<1,1,2,2>", "conv_fewout", "wino_gemm",
    "conv_wgrad<2,5>", "conv_wgrad<1,5>", "conv_wgrad<2,3>", "conv_wgrad<1,3>", "conv_wgrad<4,1>", "conv_wgrad<1,1>", "wgrad_smallk", "trunk_layer",
    "norm_fwd", "norm_bwd", "act_fwd", "act_bwd", "pack", "bias_grad", "loss", "adam", "elementwise", "sgemm"};
}

void mcvc_trace_begin_(int kind, hipStream_t s, double flops, double bytes)
{
    Rec r{kind, get_event(), get_event(), flops, bytes};
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
}
void mcvc_trace_end_(hipStream_t s) { (void)hipEventRecord(g_recs.back().b, s); }

extern "C" {
int mcvc_trace_enable(int on)
{
    g_mcvc_trace_on = on != 0;
    return 0;
}
int mcvc_trace_kinds(void) { return K_COUNT; }
const char* mcvc_trace_kind_name(int kind) { return (kind >= 0 && kind < K_COUNT) ? kNames[kind] : ""; }
// out[kind][4] = {launches, total_ms, total_flops, total_bytes}; clears the log.  Synchronises the device.
int mcvc_trace_collect(double* out)
{
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    memset(out, 0, sizeof(double) * 4 * K_COUNT);
    for (Rec& r : g_recs) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        double* o = out + 4 * r.kind;
        o[0] += 1.0; o[1] += ms; o[2] += r.flops; o[3] += r.bytes;
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    g_recs.clear();
    return 0;
}
// raw per-launch records in launch order: out[i][4] = {kind, ms, flops, bytes}; returns the count (<= max_records)
int mcvc_trace_collect_raw(double* out, int max_records)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    int n = 0;
    for (Rec& r : g_recs) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        if (n < max_records) { double* o = out + 4 * n; o[0] = r.kind; o[1] = ms; o[2] = r.flops; o[3] = r.bytes; ++n; }
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    g_recs.clear();
    return n;
}
}
