// d3m_tail.h -- hand-offs between workgroups of ONE launch: the sc1 accessors of a "last arriver" follow-up.
//
// With a step's kernels on one stream every launch is on the step's critical path, and a kernel with one round of work for
// a single workgroup behind a pass that fills the chip costs what an empty launch costs (~5 us in a replayed graph).  Where
// the follow-up is a REDUCTION it needs no launch: every workgroup publishes its partial result, takes a ticket, and the
// workgroup whose ticket is the last one adds the partials up -- nobody WAITS for anybody (d3m_lit.h fit_finish_tail: the
// objective's finish; two levels of tickets so that the adding is shared).
//
// Visibility (MI355X_MICROARCH.md, "inter-workgroup visibility": per-XCD L2s are not coherent, a CU's L1 is never
// refreshed): what the last arriver READS of what other workgroups of the same launch wrote must have been written with
// agent-scope atomics or sc1 (write-through) stores -- tail_store() --, behind which the writing wave waits (s_waitcnt
// vmcnt(0)) before its lane takes the ticket (an agent-scope atomic), and must be read with sc1 loads -- tail_load().  That is
// the guide's "valid forms" row 1; no release / acquire fence is paid for a few words per workgroup.
//
// WHAT DOES NOT WORK (round 5, measured): letting the last K arrivers WAIT (spin on the ticket word) until every workgroup
// of the launch has arrived, and then share a follow-up that is too big for one workgroup -- the tiles' slice allocation
// behind the binning pass's counts, the plan's allocations, the per-pixel passes of the faces a gathered pass set aside.
// The argument for it -- a workgroup takes its ticket only when its own work is done, so the few that wait hold a few of the
// chip's thousands of workgroup slots while all others run to completion -- assumes that the hardware dispatches the
// launch's remaining workgroups to ANY free slot.  It does not: with 1024-thread workgroups (two slots per CU) the
// 3136-workgroup counting pass of the 32-view headline, 49 of its workgroups waiting, took 2.7 s, 7.7 s and 57 s in three
// consecutive launches (0.08 ms without the wait) -- the dispatcher stalls behind CUs whose slots the waiting workgroups
// hold until something (a queue time slice) moves them; the same launch shape with 256-thread workgroups (eight slots per
// CU) ran through, i.e. would have stalled the day eight waiters met on one CU.  A workgroup must never wait for a
// workgroup that may not have been dispatched yet: HIP promises nothing about dispatch (the guide says so), and this is
// what that means in practice.  Those follow-ups stay launches of their own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace d3m {

// sc1 (agent-scope, write-through / L1-bypassing) accesses of the hand-off payloads
template <class T>
__device__ __forceinline__ void tail_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ T tail_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tail_store4(float4* p, float4 v) {
    float* f = reinterpret_cast<float*>(p);
    tail_store(f + 0, v.x); tail_store(f + 1, v.y); tail_store(f + 2, v.z); tail_store(f + 3, v.w);
}
__device__ __forceinline__ float4 tail_load4(const float4* p) {
    const float* f = reinterpret_cast<const float*>(p);
    return make_float4(tail_load(f + 0), tail_load(f + 1), tail_load(f + 2), tail_load(f + 3));
}

}  // namespace d3m
