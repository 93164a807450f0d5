// Grouped ("twin") launches: two networks of the same architecture in ONE grid.
//
// The step runs G_A2B and G_B2A (and the discriminator pairs) through identical layer schedules on different weights
// (reference train.py:203-210, 255-273).  At one or two samples per pass every kernel of such a schedule under-fills 256 CUs, and
// running the two schedules on two HIP streams costs twice the dispatches for kernels that then compete for the same hardware queues.
// Here every kernel takes a two-entry argument table and picks its entry by blockIdx.z: a plain launch has gridDim.z == 1, a grouped
// launch gridDim.z == 2 -- half the launches, twice the workgroups per launch, both chains in lock-step by construction.
//
// Host side: the layer schedules (net.hip) are written for ONE network.  A grouped pass walks the schedule twice on the host:
//   phase 1 -- with network 0's pointers: every launch is RECORDED (kernel, grid, arguments), nothing reaches the stream, stream
//              operations (event record / wait) are skipped;
//   phase 2 -- with network 1's pointers: every launch is paired with its record (same kernel and grid or the pass fails with
//              MCVC_ERR_INVALID -- the two walks take their decisions from the shapes, which are equal) and goes out ONCE with both
//              argument sets; stream operations are issued.
// The two sets may differ in pointers only (weights, activations, workspaces, destinations).
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#include <vector>

struct TwinRec { const void* fn; unsigned gx, gy, bx; size_t lds, arg_off, arg_size; };
struct TwinCtx {
    int phase = 0;
    size_t next = 0;
    int err = 0;
    std::vector<TwinRec> recs;
    std::vector<char> args;
};
extern thread_local TwinCtx* g_mcvc_twin;
static inline int mcvc_twin_phase() { return g_mcvc_twin ? g_mcvc_twin->phase : 0; }

// stream operations of a pass: issued once per grouped pass (phase 2), never while recording
static inline int mcvc_event_record(hipEvent_t e, hipStream_t s) { return mcvc_twin_phase() == 1 ? 0 : (int)hipEventRecord(e, s); }
static inline int mcvc_stream_wait(hipStream_t s, hipEvent_t e) { return mcvc_twin_phase() == 1 ? 0 : (int)hipStreamWaitEvent(s, e, 0); }
