// mcvc_launch: the one way kernels of this library are launched (see twin.h for the grouped-launch scheme).
#pragma once
#include "twin.h"

template <class A> struct Twin { A v[2]; };

// every kernel of the library is launched through this (kernels take `const Twin<A>` and read `tw.v[blockIdx.z]`)
template <class A>
static inline void mcvc_launch(void (*kern)(const Twin<A>), dim3 grid, dim3 block, size_t lds, hipStream_t s, const A& a)
{
    TwinCtx* t = g_mcvc_twin;
    Twin<A> tw;
    if (!t || t->phase == 0) {
        tw.v[0] = a; tw.v[1] = a;
        hipLaunchKernelGGL(kern, grid, block, lds, s, tw);
        return;
    }
    if (grid.z != 1 || block.y != 1 || block.z != 1) { t->err = 1001; return; }
    if (t->phase == 1) {
        TwinRec r{reinterpret_cast<const void*>(kern), grid.x, grid.y, block.x, lds, t->args.size(), sizeof(A)};
        t->args.resize(t->args.size() + sizeof(A));
        memcpy(t->args.data() + r.arg_off, &a, sizeof(A));
        t->recs.push_back(r);
        return;
    }
    if (t->next >= t->recs.size()) { t->err = 1001; return; }
    const TwinRec& r = t->recs[t->next++];
    if (r.fn != reinterpret_cast<const void*>(kern) || r.gx != grid.x || r.gy != grid.y || r.bx != block.x || r.lds != lds || r.arg_size != sizeof(A)) {
        t->err = 1001;
        return;
    }
    memcpy(&tw.v[0], t->args.data() + r.arg_off, sizeof(A));
    tw.v[1] = a;
    grid.z = 2;
    hipLaunchKernelGGL(kern, grid, block, lds, s, tw);
}

