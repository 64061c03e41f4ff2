/*
 * ftn_trace4.hip -- the production traversal kernels: Scene::intersect / Scene::intersect_test (src/bvh.rs:160-266) over 128-byte
 * four-box records (DScene::quad, built by build_quads in ftn_host.cpp: the reference's tree, two levels per record).
 *
 * Why this shape on gfx950: the reference's 32-byte nodes are gathered at random places of a 640 MB array, and the memory system
 * moves 128-byte lines -- a node visit costs a line (from L2, or from HBM) and a dependent round trip for ONE box test.  A four-box
 * record is exactly one line: one round trip decides the next two levels, boxes of children the ray does not enter are never fetched
 * on their own, and a ray makes ~2.6x fewer dependent fetches than node visits of the reference walk (measured).
 *
 * Exactness (closest hits are bit-identical to the reference walk's, tests/test_quad_bvh.py and the GPU parity suite):
 *   1. Skipping the two intermediate boxes is exact: Bounds3f::intersect_test (bounds.rs:214-233) is monotone in the box, so a ray
 *      that enters a grandchild's box enters the child's box for the same or any larger t_max (see build_quads).
 *   2. Order: the reference visits near-child-first by dir_is_neg[split_axis] at every node (bvh.rs:187-196).  For the four
 *      grandchildren [A0, A1, B0, B1] of R that order follows from the axes of R, A and B carried in the record.
 *   3. Deferred children: the reference tests a far child's box when it is POPPED, against the t_max of that moment.  Its test is
 *      `!(t0 > min(t_max, exits))` with t0 / exits independent of t_max; here a child that passes when its record is visited is
 *      pushed with its entry distance t0 (so t0 <= exits is known) and re-tested at pop time as `!(t0 > t_max)`: the same boolean.
 *      A child that fails when its record is visited also fails later (t_max only shrinks).
 *   4. Leaves: primitives in order, `t == t_max` accepted, exactly as before (prim_hit).
 *   5. The exception to 1: with a direction component of exactly zero (1/d infinite) and the origin exactly on a bounding plane the
 *      reference's test multiplies 0 by infinity, and the resulting NaN drops a constraint from the child's test that the parent's
 *      test still has -- a child can pass where its parent fails (tests/test_quad_bvh.py has the cases).  Rays with a non-finite 1/d
 *      component are therefore not walked here at all: the kernel appends them to an exception queue (WfBuffers::q_exc_*) and the
 *      host launches the reference-order kernel (k_wf_trace, ftn_wavefront.hip) over that queue right behind this one.  Jittered
 *      camera rays, BSDF-sampled and light-sampled directions practically never qualify; hand-made axis-parallel ray batches do.
 * Any-hit rays (k_wf_trace4_any): t_max never shrinks and only the boolean matters, so order is free and entries need no t0.
 *
 * Per-lane stack: LDS, [level][lane], 8-byte (closest: link, t0) or 4-byte (any-hit) entries.  LDS holds the first `lds_entries`
 * levels; deeper pushes (rare: the host sizes LDS for the occupancy it wants, DScene::quad_stack_bound is the true bound) go to a
 * per-lane spill area in global memory.
 * Scheduling is the one of k_wf_trace: persistent workgroups, per-wave chunks of the queue (one slice per XCD), idle lanes re-armed by
 * ballot ranks, node steps and leaf steps as separate convergent bodies.
 */
#include "ftn_wf_common.h"

namespace ftn {

enum : uint32_t { T4_IDLE = 0, T4_NODE = 1, T4_LEAF = 2 };
#define T4_NONE 0xffffffffu      /* no entry (the link word of an empty slot; a child the ray does not enter) */

/* Bounds3f::intersect_test on a stored slot (a = {min.x, max.x, min.y, max.y}, b = {min.z, max.z, link, meta}): pass / fail, the
 * clipped entry distance t0 = max(0, nears) -- the value the reference's test of this box compares against min(t_max, exits) -- and
 * t1 = min(t_max, exits).  The reference's `if t_near > t_far { swap }` is written as (min, max) of the two plane distances: the same
 * two values whenever neither is a NaN, and the rays whose products can be NaN (0 * inf) never get here (ray_is_exceptional). */
typedef float t4_f2 __attribute__((ext_vector_type(2)));
__device__ inline bool slab4(float4 a, float4 b, V3 o, V3 inv, float t_max, float* t0o, float* t1o) {
    const float k = 1.0f + 2.0f * gamma_n(3);
    /* the (min, max) pair of an axis sits in an even / odd register pair after the dwordx4 loads: packed f32 subtract and multiply
     * (v_pk_add_f32 / v_pk_mul_f32; each half is the plain IEEE operation) */
    const t4_f2 px = {a.x, a.y}, py = {a.z, a.w}, pz = {b.x, b.y};
    const t4_f2 tx = (px - (t4_f2){o.x, o.x}) * (t4_f2){inv.x, inv.x};
    const t4_f2 ty = (py - (t4_f2){o.y, o.y}) * (t4_f2){inv.y, inv.y};
    const t4_f2 tz = (pz - (t4_f2){o.z, o.z}) * (t4_f2){inv.z, inv.z};
    /* the reference scales every axis' far distance by k = 1 + 2 gamma(3) before taking the minimum; x -> fl(x * k) is non-decreasing
     * (k > 0, rounding is monotone), so it commutes with min: one multiplication of the smallest far distance gives the same bits */
    const float far_k = fmin_(fmin_(fmax_(tx.x, tx.y), fmax_(ty.x, ty.y)), fmax_(tz.x, tz.y)) * k;
    const float t0 = fmax_(fmax_(fmax_(0.0f, fmin_(tx.x, tx.y)), fmin_(ty.x, ty.y)), fmin_(tz.x, tz.y));
    const float t1 = fmin_(t_max, far_k);
    *t0o = t0; *t1o = t1;
    return !(t0 > t1);
}

/* queue bookkeeping shared by both kernels: the wave's current chunk, its XCD slice */
struct WaveQueue {
    uint32_t chunk_next, chunk_end, slice, slices_done, chunk; bool exhausted;
};
__device__ inline void wq_init(WaveQueue& q, uint32_t count, uint32_t chunk_max) {
    q.chunk_next = 0; q.chunk_end = 0; q.exhausted = count == 0; q.slice = blockIdx.x & 7u; q.slices_done = 0;
    uint32_t c = count / (gridDim.x * 4u * 16u); c &= ~63u; q.chunk = c < 64u ? 64u : (c > chunk_max ? chunk_max : c);
}
/* makes sure the wave holds a non-empty chunk (or marks the queue exhausted); returns true at the moment the wave finds every slice dry */
__device__ inline bool wq_refill(WaveQueue& q, uint32_t count, uint32_t* head, uint32_t lane) {
    bool just_dry = false;
    while (q.chunk_next == q.chunk_end && !q.exhausted) {
        const uint32_t s_lo = (uint32_t)(((unsigned long long)count * q.slice) >> 3) & ~63u, s_hi = q.slice == 7u ? count : ((uint32_t)(((unsigned long long)count * (q.slice + 1u)) >> 3) & ~63u);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(head + CTR(q.slice), q.chunk);
        base = __shfl(base, 0, 64) + s_lo;
        if (base >= s_hi) { q.slice = (q.slice + 1u) & 7u; if (++q.slices_done == 8u) { q.exhausted = true; just_dry = true; } }
        else { q.chunk_next = base; q.chunk_end = base + q.chunk < s_hi ? base + q.chunk : s_hi; }
    }
    return just_dry;
}

/* per-ray constants of the slab test and of Triangle::intersect (triangle.rs:189-205) */
struct RaySetup { V3 o, inv, dorig; float t_max, sx, sy, sz; uint32_t neg24 /* dir_is_neg bits, repeated in bytes 0, 1, 2: lines up with the record's three one-hot split axes */; int kz; };
/* a component of 1/d that is infinite or NaN: point 5 of the header comment.  Also a ray whose plane distances overflow (|o * 1/d| infinite:
 * a denormal direction component next to an astronomic origin): an empty slot's box {x, y in [0, 0], z in [inf, inf]} fails the slab test
 * for every other ray -- for such a ray with t_max infinite both of its x / y distances can be +inf and the test would pass. */
/* (by value: read through a reference, the loads that feed the three products are merged into overlapping two-float loads before this body
 * is inlined, and the ray constants then live in scratch memory in the kernels) */
__device__ inline bool ray_is_exceptional(float ox, float oy, float oz, float ix, float iy, float iz) {
    return !(fabsf(ix) < FTN_INF && fabsf(iy) < FTN_INF && fabsf(iz) < FTN_INF && ix != 0.0f && iy != 0.0f && iz != 0.0f &&
             fabsf(ox) < FTN_INF && fabsf(oy) < FTN_INF && fabsf(oz) < FTN_INF &&
             fabsf(ox * ix) < FTN_INF && fabsf(oy * iy) < FTN_INF && fabsf(oz * iz) < FTN_INF);
}
template <bool SPHERES>
__device__ inline void ray_setup(RaySetup& R, float4 a, float4 b) {
    R.o = V3(a.x, a.y, a.z); const V3 d(b.x, b.y, b.z); R.t_max = b.w;
    if (SPHERES) R.dorig = d;
    R.inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    R.neg24 = ((d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u)) * 0x010101u;
    R.kz = max_dimension(vabs(d));
    /* permute_vector(d, kx, ky, kz) with (kx, ky, kz) = (kz + 1, kz + 2, kz) mod 3, as selects (no dynamic indexing: that would put the vector in scratch) */
    const bool k0 = R.kz == 0, k1 = R.kz == 1;
    const V3 dperm(k0 ? d.y : (k1 ? d.z : d.x), k0 ? d.z : (k1 ? d.x : d.y), k0 ? d.x : (k1 ? d.y : d.z));
    R.sx = -dperm.x / dperm.z; R.sy = -dperm.y / dperm.z; R.sz = 1.0f / dperm.z;
}

/* ------------------------------------------------------------------ per-lane stack: LDS levels first, then the global spill area
 * CHECK = false: the caller has made sure (one ballot per step) that every lane of the wave stays inside LDS during this step */
template <class E> struct T4Stack {
    E* base; E* lim; E* sp; E* spill; size_t n_lanes, gl; uint32_t spill_levels;
    /* (the level is clamped to the spill area: DScene::quad_stack_bound makes a deeper level impossible, and an out-of-bounds store on
     * a shared GPU is not an acceptable way to find out otherwise) */
    __device__ inline size_t spill_at() const { const uint32_t lv = (uint32_t)((sp - lim) >> 8); return (size_t)(lv < spill_levels ? lv : spill_levels - 1u) * n_lanes + gl; }
    template <bool CHECK> __device__ inline void push(E v) {
        if (!CHECK || sp < lim) *sp = v; else if (spill_levels) spill[spill_at()] = v;
        sp += 256;
    }
    template <bool CHECK> __device__ inline E pop() {
        sp -= 256;
        if (!CHECK || sp < lim) return *sp;
        return spill_levels ? spill[spill_at()] : *base;
    }
    __device__ inline bool empty() const { return sp == base; }
};

/* ------------------------------------------------------------------ closest hit */
struct T4Lane { uint32_t mode, cur, lp; bool finish; };

/* The reference's test of a deferred child at the moment it is popped (bvh.rs:173 with today's t_max) is !(t0 > t_max): entries that
 * fail it are dropped, the first one that passes is visited. */
template <bool CHECK>
__device__ inline void t4_pop_next(T4Stack<uint2>& St, T4Lane& L, float t_max) {
    for (;;) {
        if (St.empty()) { L.finish = true; return; }
        const uint2 v = St.template pop<CHECK>();
        if (!(__uint_as_float(v.y) > t_max)) {
            if (v.x >> 31) { L.lp = v.x & 0x7fffffffu; L.mode = T4_LEAF; } else { L.cur = v.x; L.mode = T4_NODE; }
            return;
        }
    }
}

/* one record: four box tests, the reference's visiting order, at most three pushes */
template <bool CHECK>
__device__ inline void t4_record_step(const DScene& S, const RaySetup& R, T4Stack<uint2>& St, T4Lane& L) {
    const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.quad) + L.cur);
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5], q6 = rec[6], q7 = rec[7];
    float ta, tb, tc, td, x1;
    const bool ha = slab4(q0, q1, R.o, R.inv, R.t_max, &ta, &x1), hb = slab4(q2, q3, R.o, R.inv, R.t_max, &tb, &x1);       /* (an empty slot holds a box no ray enters) */
    const bool hc = slab4(q4, q5, R.o, R.inv, R.t_max, &tc, &x1), hd = slab4(q6, q7, R.o, R.inv, R.t_max, &td, &x1);
    /* entries as stored: byte offset of an interior child's record, or first primitive | bit 31 for a leaf; a child the ray does not
     * enter becomes T4_NONE here, so that "entered" travels through the reordering inside the entry word */
    const uint32_t ea = ha ? __float_as_uint(q1.z) : T4_NONE, eb = hb ? __float_as_uint(q3.z) : T4_NONE, ec = hc ? __float_as_uint(q5.z) : T4_NONE, ed = hd ? __float_as_uint(q7.z) : T4_NONE;
    /* the reference's order: pair A = slots (a, b) ordered by A's axis, pair B = (c, d) by B's axis, the pairs by R's axis.  Slot a's
     * meta word holds the three one-hot axes in bytes 0 (A), 1 (R), 2 (B); neg24 holds dir_is_neg in the same three bytes */
    const uint32_t ax = __float_as_uint(q1.w) & R.neg24;
    const bool sA = (ax & 0xffu) != 0, sR = (ax & 0xff00u) != 0, sB = (ax & 0xff0000u) != 0;
    const uint32_t ex0 = sA ? eb : ea, ex1 = sA ? ea : eb, ey0 = sB ? ed : ec, ey1 = sB ? ec : ed;
    const float tx0 = sA ? tb : ta, tx1 = sA ? ta : tb, ty0 = sB ? td : tc, ty1 = sB ? tc : td;
    const uint32_t e0 = sR ? ey0 : ex0, e1 = sR ? ey1 : ex1, e2 = sR ? ex0 : ey0, e3 = sR ? ex1 : ey1;
    const float t1_ = sR ? ty1 : tx1, t2_ = sR ? tx0 : ty0, t3_ = sR ? tx1 : ty1;
    const bool h0 = e0 != T4_NONE, h1 = e1 != T4_NONE, h2 = e2 != T4_NONE, h3 = e3 != T4_NONE;
    /* the first child in order that the ray enters is visited next; the later ones wait on the stack with their t0 */
    const bool h01 = h0 || h1, h012 = h01 || h2;
    if (h3 && h012) St.template push<CHECK>(make_uint2(e3, __float_as_uint(t3_)));
    if (h2 && h01) St.template push<CHECK>(make_uint2(e2, __float_as_uint(t2_)));
    if (h1 && h0) St.template push<CHECK>(make_uint2(e1, __float_as_uint(t1_)));
    if (h012 || h3) {
        const uint32_t next = h0 ? e0 : (h1 ? e1 : (h2 ? e2 : e3));
        if (next >> 31) { L.lp = next & 0x7fffffffu; L.mode = T4_LEAF; } else L.cur = next;
    } else t4_pop_next<CHECK>(St, L, R.t_max);
}

template <bool COUNT, bool SPHERES, int BURST>
__global__ void __launch_bounds__(256, (SPHERES ? 1 : 5)) k_wf_trace4(DScene S, WfBuffers W, const uint32_t* __restrict__ queue, const uint32_t* count_ptr, uint32_t* head, DevStats* stats,
                                                   uint32_t refill, uint32_t leaf_batch, uint32_t chunk_max, uint32_t lds_entries, uint2* __restrict__ spill, uint32_t spill_levels) {
    T4Stack<uint2> St;
    extern __shared__ uint2 lds_stack2[];
    St.base = lds_stack2 + threadIdx.x;                           /* [level][lane]: ds_write_b64 / ds_read_b64, conflict-free */
    St.lim = St.base + 256u * lds_entries; St.sp = St.base; St.spill = spill; St.spill_levels = spill_levels;
    St.n_lanes = (size_t)gridDim.x * 256u; St.gl = (size_t)blockIdx.x * 256u + threadIdx.x;
    /* a record step pushes at most three entries: above st_soft a step takes the bounds-checked stack operations.  (LDS addresses are
     * 32-bit and unsigned: `lim - 3 levels` must not wrap below the first level, so a stack of fewer than three LDS levels is always checked) */
    const bool st_tiny = lds_entries < 3u;
    uint2* const st_soft = st_tiny ? St.base : St.lim - 3 * 256;
    const uint32_t count = *count_ptr;
    const uint32_t lane = lane_id();
    unsigned long long n_rec = 0, n_prim = 0;
    unsigned long long occ[7] = {0, 0, 0, 0, 0, 0, 0};       /* COUNT: DevStats::t4_occ (wave-uniform values, added by lane 0) */
    T4Lane L; L.mode = T4_IDLE; L.cur = 0; L.lp = 0; L.finish = false;
    WaveQueue Q; wq_init(Q, count, chunk_max);
    uint32_t rid = 0;
    RaySetup R; R.o = V3(0.0f, 0.0f, 0.0f); R.inv = R.o; R.dorig = R.o; R.t_max = 0.0f; R.neg24 = 0; R.kz = 0; R.sx = R.sy = R.sz = 0.0f;
    int hprim = -1; float hb0 = 0.0f, hb1 = 0.0f, hb2 = 0.0f; bool found = false;
    const float4 rlo = make_float4(S.root_lo[0], S.root_lo[1], S.root_lo[2], 0.0f), rhi = make_float4(S.root_hi[0], S.root_hi[1], S.root_hi[2], 0.0f);
    for (;;) {
        /* ---- re-arm idle lanes */
        const unsigned long long idle = __ballot(L.mode == T4_IDLE);
        if (!Q.exhausted && (uint32_t)__popcll(idle) >= refill) {
            const uint32_t need = (uint32_t)__popcll(idle);
            if (wq_refill(Q, count, head, lane)) {
                /* the launch starts to drain: let the stream that waits for it (the any-hit trace) go ahead */
                if (W.drain_sig && lane == 0 && atomicAdd(&W.counters[CTR(11)], 1u) + 1u == W.drain_at)
                    __hip_atomic_store(W.drain_sig, W.drain_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);          /* a timing hint only: no data is handed over */
            }
            const uint32_t avail = Q.chunk_end - Q.chunk_next;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            bool hand_back = false; uint32_t q_entry = 0;
            if (COUNT) { occ[5]++; occ[6] += need < avail ? need : avail; }
            if (L.mode == T4_IDLE && rank < avail) {
                float4 a, b;
                q_entry = queue[Q.chunk_next + rank];
                load_queued_ray<false>(W, q_entry, &a, &b, &rid);
                ray_setup<SPHERES>(R, a, b);
                St.sp = St.base; L.cur = 0; found = false; hprim = -1; hb0 = 0.0f; hb1 = 0.0f; hb2 = 0.0f;
                if (ray_is_exceptional(R.o.x, R.o.y, R.o.z, R.inv.x, R.inv.y, R.inv.z)) hand_back = true;            /* walked by the reference-order kernel instead */
                /* the root's own box test (bvh.rs:173 at node 0) */
                else if (S.n_nodes == 0 || !slab_test(rlo, rhi, R.o, R.inv, R.t_max)) { W.hit[rid] = make_float4(FTN_INF, 0.0f, 0.0f, 0.0f); W.hit_prim[rid] = -1; }
                else if (S.root_is_leaf) { L.lp = 0; L.mode = T4_LEAF; }
                else L.mode = T4_NODE;
            }
            if (__ballot(hand_back) != 0) wave_push(hand_back, q_entry, W.q_exc_closest, &W.counters[CTR(32)]);
            Q.chunk_next += (need < avail ? need : avail);
        }
        const unsigned long long m_node = __ballot(L.mode == T4_NODE), m_leaf = __ballot(L.mode == T4_LEAF);
        if ((m_node | m_leaf) == 0) { if (Q.exhausted) break; else continue; }
        L.finish = false;
        if (COUNT) occ[0]++;
        if (m_node != 0 && (uint32_t)__popcll(m_leaf) < leaf_batch) {
            /* ---- record steps.  One ballot per step decides whether any lane could leave LDS during it (a step pushes at most three
             * entries): nearly never, and then its pushes and pops need no bounds checks */
#pragma unroll
            for (int burst = 0; burst < BURST; burst++) {
                const bool on = L.mode == T4_NODE && !L.finish;
                if (COUNT && on) n_rec++;
                if (COUNT) { const unsigned long long m_on = __ballot(on); if (m_on) { occ[1]++; occ[2] += (unsigned long long)__popcll(m_on); } }
                if (__builtin_expect(__ballot(on && (st_tiny || St.sp > st_soft)) == 0, 1)) { if (on) t4_record_step<false>(S, R, St, L); }
                else if (on) t4_record_step<true>(S, R, St, L);
            }
        } else {
            /* ---- leaf step: one primitive per lane */
            if (COUNT) { occ[3]++; occ[4] += (unsigned long long)__popcll(m_leaf); }
            if (L.mode == T4_LEAF) {
                const uint32_t prim = L.lp;
                float4 g0 = S.geom[S.geom_stride * prim], g1 = S.geom[S.geom_stride * prim + 1], g2 = S.geom[S.geom_stride * prim + 2];
                pin4(g0); pin4(g1); pin4(g2);
                if (COUNT) n_prim++;
                float t = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
                const bool hh = prim_hit<SPHERES>(S, prim, g0, g1, g2, R.o, R.dorig, R.t_max, R.kz, R.sx, R.sy, R.sz, &t, &b0, &b1, &b2);
                if (hh) { found = true; R.t_max = t; hprim = (int)prim; hb0 = b0; hb1 = b1; hb2 = b2; }
                if (__float_as_uint(g0.w) & GF_LEAF_END) t4_pop_next<true>(St, L, R.t_max);
                else L.lp++;
            }
        }
        if (L.finish) {
            W.hit[rid] = make_float4(found ? R.t_max : FTN_INF, hb0, hb1, hb2); W.hit_prim[rid] = hprim;
            L.mode = T4_IDLE;
        }
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { n_rec += __shfl_down(n_rec, off, 64); n_prim += __shfl_down(n_prim, off, 64); }
        if (lane == 0) { if (n_rec) atomicAdd(&stats->quad_records, n_rec); if (n_prim) atomicAdd(&stats->prims_tested, n_prim); for (int k = 0; k < 7; k++) if (occ[k]) atomicAdd(&stats->t4_occ[k], occ[k]); }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats->rays_closest, (unsigned long long)count);
}

extern __shared__ uint32_t lds_stack1[];
/* ------------------------------------------------------------------ any hit
 * Order is free (the boolean does not depend on it).  POLICY 1: of the children the ray enters, the one whose box it stays in longest
 * is walked first -- the policy that ended blocked rays soonest in the two-box kernel; POLICY 0: slot
 * order (fewer instructions per step). */
template <bool CHECK, int POLICY>
__device__ inline void t4_any_step(const DScene& S, const RaySetup& R, T4Stack<uint32_t>& St, T4Lane& L) {
    const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.quad) + L.cur);
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5], q6 = rec[6], q7 = rec[7];
    float a0, a1, b0, b1, c0, c1, d0, d1;
    const bool ha = slab4(q0, q1, R.o, R.inv, R.t_max, &a0, &a1), hb = slab4(q2, q3, R.o, R.inv, R.t_max, &b0, &b1);
    const bool hc = slab4(q4, q5, R.o, R.inv, R.t_max, &c0, &c1), hd = slab4(q6, q7, R.o, R.inv, R.t_max, &d0, &d1);
    const uint32_t ea = __float_as_uint(q1.z), eb = __float_as_uint(q3.z), ec = __float_as_uint(q5.z), ed = __float_as_uint(q7.z);
    uint32_t next = 0; bool have = ha || hb || hc || hd;
    if (POLICY == 1) {
        /* length of the ray's stay in each entered box, -1: not entered (fmaxf drops the NaN of inf - inf) */
        const float sa = ha ? fmaxf(a1 - a0, 0.0f) : -1.0f, sb = hb ? fmaxf(b1 - b0, 0.0f) : -1.0f, sc = hc ? fmaxf(c1 - c0, 0.0f) : -1.0f, sd = hd ? fmaxf(d1 - d0, 0.0f) : -1.0f;
        const float best = fmaxf(fmaxf(sa, sb), fmaxf(sc, sd));
        const bool pa = sa == best, pb = !pa && sb == best, pc = !pa && !pb && sc == best, pd = !pa && !pb && !pc;
        next = pa ? ea : (pb ? eb : (pc ? ec : ed));
        if (ha && !pa) St.template push<CHECK>(ea);
        if (hb && !pb) St.template push<CHECK>(eb);
        if (hc && !pc) St.template push<CHECK>(ec);
        if (hd && !pd) St.template push<CHECK>(ed);
    } else {
        next = ha ? ea : (hb ? eb : (hc ? ec : ed));
        if (hd && (ha || hb || hc)) St.template push<CHECK>(ed);
        if (hc && (ha || hb)) St.template push<CHECK>(ec);
        if (hb && ha) St.template push<CHECK>(eb);
    }
    if (!have) {
        if (St.empty()) { L.finish = true; return; }
        next = St.template pop<CHECK>();
    }
    if (next >> 31) { L.lp = next & 0x7fffffffu; L.mode = T4_LEAF; } else L.cur = next;
}

template <bool COUNT, bool SPHERES, int BURST, int POLICY>
__global__ void __launch_bounds__(256) k_wf_trace4_any(DScene S, WfBuffers W, const uint32_t* __restrict__ queue, const uint32_t* count_ptr, uint32_t* head, DevStats* stats,
                                                       uint32_t refill, uint32_t leaf_batch, uint32_t chunk_max, uint32_t lds_entries, uint32_t* __restrict__ spill, uint32_t spill_levels) {
    T4Stack<uint32_t> St;
    St.base = lds_stack1 + threadIdx.x;
    St.lim = St.base + 256u * lds_entries; St.sp = St.base; St.spill = spill; St.spill_levels = spill_levels;
    St.n_lanes = (size_t)gridDim.x * 256u; St.gl = (size_t)blockIdx.x * 256u + threadIdx.x;
    const bool st_tiny = lds_entries < 3u;                         /* see k_wf_trace4 */
    uint32_t* const st_soft = st_tiny ? St.base : St.lim - 3 * 256;
    const uint32_t count = *count_ptr;
    const uint32_t lane = lane_id();
    unsigned long long n_rec = 0, n_prim = 0;
    unsigned long long occ[7] = {0, 0, 0, 0, 0, 0, 0};
    T4Lane L; L.mode = T4_IDLE; L.cur = 0; L.lp = 0; L.finish = false;
    WaveQueue Q; wq_init(Q, count, chunk_max);
    uint32_t rid = 0;
    RaySetup R; R.o = V3(0.0f, 0.0f, 0.0f); R.inv = R.o; R.dorig = R.o; R.t_max = 0.0f; R.neg24 = 0; R.kz = 0; R.sx = R.sy = R.sz = 0.0f;
    bool found = false;
    const float4 rlo = make_float4(S.root_lo[0], S.root_lo[1], S.root_lo[2], 0.0f), rhi = make_float4(S.root_hi[0], S.root_hi[1], S.root_hi[2], 0.0f);
    for (;;) {
        const unsigned long long idle = __ballot(L.mode == T4_IDLE);
        if (!Q.exhausted && (uint32_t)__popcll(idle) >= refill) {
            const uint32_t need = (uint32_t)__popcll(idle);
            (void)wq_refill(Q, count, head, lane);
            const uint32_t avail = Q.chunk_end - Q.chunk_next;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            bool hand_back = false; uint32_t q_entry = 0;
            if (COUNT) { occ[5]++; occ[6] += need < avail ? need : avail; }
            if (L.mode == T4_IDLE && rank < avail) {
                float4 a, b;
                q_entry = queue[Q.chunk_next + rank];
                load_queued_ray<true>(W, q_entry, &a, &b, &rid);
                ray_setup<SPHERES>(R, a, b);
                St.sp = St.base; L.cur = 0; found = false;
                if (ray_is_exceptional(R.o.x, R.o.y, R.o.z, R.inv.x, R.inv.y, R.inv.z)) hand_back = true;
                else if (S.n_nodes == 0 || !slab_test(rlo, rhi, R.o, R.inv, R.t_max)) store_occluded(W, rid, false);
                else if (S.root_is_leaf) { L.lp = 0; L.mode = T4_LEAF; }
                else L.mode = T4_NODE;
            }
            if (__ballot(hand_back) != 0) wave_push(hand_back, q_entry, W.q_exc_any, &W.counters[CTR(33)]);
            Q.chunk_next += (need < avail ? need : avail);
        }
        const unsigned long long m_node = __ballot(L.mode == T4_NODE), m_leaf = __ballot(L.mode == T4_LEAF);
        if ((m_node | m_leaf) == 0) { if (Q.exhausted) break; else continue; }
        L.finish = false;
        if (COUNT) occ[0]++;
        if (m_node != 0 && (uint32_t)__popcll(m_leaf) < leaf_batch) {
#pragma unroll
            for (int burst = 0; burst < BURST; burst++) {
                const bool on = L.mode == T4_NODE && !L.finish;
                if (COUNT && on) n_rec++;
                if (COUNT) { const unsigned long long m_on = __ballot(on); if (m_on) { occ[1]++; occ[2] += (unsigned long long)__popcll(m_on); } }
                if (__builtin_expect(__ballot(on && (st_tiny || St.sp > st_soft)) == 0, 1)) { if (on) t4_any_step<false, POLICY>(S, R, St, L); }
                else if (on) t4_any_step<true, POLICY>(S, R, St, L);
            }
        } else {
            if (COUNT) { occ[3]++; occ[4] += (unsigned long long)__popcll(m_leaf); }
            if (L.mode == T4_LEAF) {
                const uint32_t prim = L.lp;
                float4 g0 = S.geom[S.geom_stride * prim], g1 = S.geom[S.geom_stride * prim + 1], g2 = S.geom[S.geom_stride * prim + 2];
                pin4(g0); pin4(g1); pin4(g2);
                if (COUNT) n_prim++;
                float t = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
                const bool hh = prim_hit<SPHERES>(S, prim, g0, g1, g2, R.o, R.dorig, R.t_max, R.kz, R.sx, R.sy, R.sz, &t, &b0, &b1, &b2);
                if (hh) { found = true; L.finish = true; }
                else if (__float_as_uint(g0.w) & GF_LEAF_END) {
                    if (St.empty()) L.finish = true;
                    else { const uint32_t next = St.template pop<true>(); if (next >> 31) L.lp = next & 0x7fffffffu; else { L.cur = next; L.mode = T4_NODE; } }
                } else L.lp++;
            }
        }
        if (L.finish) { store_occluded(W, rid, found); L.mode = T4_IDLE; }
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { n_rec += __shfl_down(n_rec, off, 64); n_prim += __shfl_down(n_prim, off, 64); }
        if (lane == 0) {
            if (n_rec) { atomicAdd(&stats->quad_records, n_rec); atomicAdd(&stats->quad_records_any, n_rec); }
            if (n_prim) { atomicAdd(&stats->prims_tested, n_prim); atomicAdd(&stats->prims_any, n_prim); }
            for (int k = 0; k < 7; k++) if (occ[k]) atomicAdd(&stats->t4_occ[7 + k], occ[k]);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats->rays_any, (unsigned long long)count);
}

/* ------------------------------------------------------------------ any hit over eight-box occlusion records (k_wf_trace8_any; DScene::oct, build_octs in ftn_host.cpp)
 * intersect_test returns a boolean and never shrinks t_max: a ray is occluded iff some LEAF's own box passes the slab test and one of the
 * leaf's primitives passes its test -- the interior boxes on the way only have to let every such leaf be reached, so they may be larger
 * than the reference's.  The records hold up to eight children with 8-bit planes on a per-record grid, rounded outwards by the builder
 * (decoded plane = origin + q * step with lo_dec <= lo and hi_dec >= hi as real numbers); the kernel's test of a decoded box is
 *     t = q * (step / d) + (origin - o) / d          one fused multiply-add per plane, near / far plane picked by the sign of d,
 * pushed outwards by m = (|(origin - o) / d| + 255 |step / d|) * 2^-20 -- four times the rounding error of this expression plus that of
 * the reference's own (lo - o) * (1 / d), so the test passes whenever the reference's test of ANY box inside the decoded one passes (the
 * far side is also scaled by 1 + 4 gamma(3) >= the reference's 1 + 2 gamma(3)).  A leaf is then tested exactly: Bounds3f::intersect_test
 * on the leaf's true box (the min / max of a single triangle's vertices, or the explicit box of `oct_xbox`), then the triangle test.
 * Rays whose 1/d or o/d are outside the range where these bounds hold (|1/d| > 2^60, |o| > 2^40) join the exceptional rays in the queue
 * of the reference-order kernel.  Same scheduling as k_wf_trace4_any: persistent workgroups, XCD queue slices, LDS stack with a global
 * spill area, record steps and leaf steps as separate convergent bodies.  ~12 records per ray instead of 20 four-box records. */
enum : uint32_t { T8_IDLE = 0, T8_NODE = 1, T8_LEAF = 2, T8_LEAF_IN = 3 /* inside a leaf of several primitives whose box has passed */ };
#define T8_XBOX 0x40000000u
__device__ inline bool ray_out_of_range8(float ox, float oy, float oz, float ix, float iy, float iz) {
    const float big_i = 1.152921504606846976e18f /* 2^60 */, big_o = 1.099511627776e12f /* 2^40 */;
    return !(fabsf(ix) <= big_i && fabsf(iy) <= big_i && fabsf(iz) <= big_i && fabsf(ox) <= big_o && fabsf(oy) <= big_o && fabsf(oz) <= big_o);
}
struct T8Axis { float a, bn, bf; uint32_t n03, n47, f03, f47; };
/* per record and axis: the two constants of the plane expression, the outward margin folded into the offsets, near / far byte rows by sign */
__device__ inline T8Axis t8_axis(float org, uint32_t ebyte, float o, float inv, uint32_t lo03, uint32_t lo47, uint32_t hi03, uint32_t hi47) {
    T8Axis A;
    const float step = __uint_as_float(ebyte << 23);
    A.a = step * inv;
    const float b = (org - o) * inv;
    const float m = (fabsf(b) + 255.0f * fabsf(A.a)) * 9.5367431640625e-07f;
    A.bn = b - m; A.bf = b + m;
    const bool neg = inv < 0.0f;
    A.n03 = neg ? hi03 : lo03; A.n47 = neg ? hi47 : lo47; A.f03 = neg ? lo03 : hi03; A.f47 = neg ? lo47 : hi47;
    return A;
}
#define T8_BYTE(w, k) ((float)(((w) >> (8 * (k))) & 0xffu))
/* child c of the record: conservative entry / exit distances.  The near and the far plane of an axis share the slope `a`: one packed
 * fused multiply-add (v_pk_fma_f32: each half is the scalar fma) gives both */
#define T8_CHILD(c, t0v, t1v) \
    { const uint32_t kx_ = (c) & 3; \
      const t4_f2 tx_ = __builtin_elementwise_fma((t4_f2){T8_BYTE((c) < 4 ? X.n03 : X.n47, kx_), T8_BYTE((c) < 4 ? X.f03 : X.f47, kx_)}, (t4_f2){X.a, X.a}, (t4_f2){X.bn, X.bf}); \
      const t4_f2 ty_ = __builtin_elementwise_fma((t4_f2){T8_BYTE((c) < 4 ? Y.n03 : Y.n47, kx_), T8_BYTE((c) < 4 ? Y.f03 : Y.f47, kx_)}, (t4_f2){Y.a, Y.a}, (t4_f2){Y.bn, Y.bf}); \
      const t4_f2 tz_ = __builtin_elementwise_fma((t4_f2){T8_BYTE((c) < 4 ? Z.n03 : Z.n47, kx_), T8_BYTE((c) < 4 ? Z.f03 : Z.f47, kx_)}, (t4_f2){Z.a, Z.a}, (t4_f2){Z.bn, Z.bf}); \
      t0v = fmax_(fmax_(fmax_(0.0f, tx_.x), ty_.x), tz_.x); t1v = fmin_(t_max, fmin_(fmin_(tx_.y, ty_.y), tz_.y) * k2); }

template <bool CHECK, int POLICY>
__device__ __forceinline__ void t8_any_step(const DScene& S, T4Stack<uint32_t>& St, T4Lane& L, V3 o, V3 inv, float t_max) {
    const uint4* rec = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(S.oct) + L.cur);
    const uint4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4], r5 = rec[5];
    const float k2 = 1.0f + 4.0f * gamma_n(3);
    const T8Axis X = t8_axis(__uint_as_float(r0.x), r0.w & 0xffu, o.x, inv.x, r3.x, r3.y, r3.z, r3.w);
    const T8Axis Y = t8_axis(__uint_as_float(r0.y), (r0.w >> 8) & 0xffu, o.y, inv.y, r4.x, r4.y, r4.z, r4.w);
    const T8Axis Z = t8_axis(__uint_as_float(r0.z), (r0.w >> 16) & 0xffu, o.z, inv.z, r5.x, r5.y, r5.z, r5.w);
    float a0, a1, b0, b1, c0, c1, d0, d1, e0, e1, f0, f1, g0, g1, h0, h1;
    T8_CHILD(0, a0, a1) T8_CHILD(1, b0, b1) T8_CHILD(2, c0, c1) T8_CHILD(3, d0, d1) T8_CHILD(4, e0, e1) T8_CHILD(5, f0, f1) T8_CHILD(6, g0, g1) T8_CHILD(7, h0, h1)
    const uint32_t la = r1.x, lb = r1.y, lc = r1.z, ld = r1.w, le = r2.x, lf = r2.y, lg = r2.z, lh = r2.w;
    const bool ha = !(a0 > a1) && la != T4_NONE, hb = !(b0 > b1) && lb != T4_NONE, hc = !(c0 > c1) && lc != T4_NONE, hd = !(d0 > d1) && ld != T4_NONE;
    const bool he = !(e0 > e1) && le != T4_NONE, hf = !(f0 > f1) && lf != T4_NONE, hg = !(g0 > g1) && lg != T4_NONE, hh = !(h0 > h1) && lh != T4_NONE;
    uint32_t next = T4_NONE;
    if (POLICY == 1) {
        /* the child whose (decoded) box the ray stays in longest first: the order that ended blocked rays soonest over four-box records */
        const float sa = ha ? a1 - a0 : -1.0f, sb = hb ? b1 - b0 : -1.0f, sc = hc ? c1 - c0 : -1.0f, sd = hd ? d1 - d0 : -1.0f;
        const float se = he ? e1 - e0 : -1.0f, sf = hf ? f1 - f0 : -1.0f, sg = hg ? g1 - g0 : -1.0f, sh = hh ? h1 - h0 : -1.0f;
        const float best = fmaxf(fmaxf(fmaxf(sa, sb), fmaxf(sc, sd)), fmaxf(fmaxf(se, sf), fmaxf(sg, sh)));
        bool taken = !(best >= 0.0f);                             /* nothing entered: nothing to take */
#define T8_PICK(hx, sx, lx) { const bool me_ = hx && !taken && sx == best; if (me_) { next = lx; taken = true; } else if (hx) St.template push<CHECK>(lx); }
        T8_PICK(ha, sa, la) T8_PICK(hb, sb, lb) T8_PICK(hc, sc, lc) T8_PICK(hd, sd, ld) T8_PICK(he, se, le) T8_PICK(hf, sf, lf) T8_PICK(hg, sg, lg) T8_PICK(hh, sh, lh)
#undef T8_PICK
    } else {
        bool taken = false;
#define T8_PICK(hx, lx) { if (hx) { if (!taken) { next = lx; taken = true; } else St.template push<CHECK>(lx); } }
        T8_PICK(ha, la) T8_PICK(hb, lb) T8_PICK(hc, lc) T8_PICK(hd, ld) T8_PICK(he, le) T8_PICK(hf, lf) T8_PICK(hg, lg) T8_PICK(hh, lh)
#undef T8_PICK
    }
    if (next == T4_NONE) {
        if (St.empty()) { L.finish = true; return; }
        next = St.template pop<CHECK>();
    }
    if (next >> 31) { L.lp = next & 0x7fffffffu; L.mode = T8_LEAF; } else L.cur = next;
}

template <bool COUNT, int BURST, int POLICY>
__global__ void __launch_bounds__(256) k_wf_trace8_any(DScene S, WfBuffers W, const uint32_t* __restrict__ queue, const uint32_t* count_ptr, uint32_t* head, DevStats* stats,
                                                       uint32_t refill, uint32_t leaf_batch, uint32_t chunk_max, uint32_t lds_entries, uint32_t* __restrict__ spill, uint32_t spill_levels) {
    T4Stack<uint32_t> St;
    St.base = lds_stack1 + threadIdx.x;
    St.lim = St.base + 256u * lds_entries; St.sp = St.base; St.spill = spill; St.spill_levels = spill_levels;
    St.n_lanes = (size_t)gridDim.x * 256u; St.gl = (size_t)blockIdx.x * 256u + threadIdx.x;
    const bool st_tiny = lds_entries < 7u;                         /* a record step pushes at most seven entries */
    uint32_t* const st_soft = st_tiny ? St.base : St.lim - 7 * 256;
    const uint32_t count = *count_ptr;
    const uint32_t lane = lane_id();
    unsigned long long n_rec = 0, n_prim = 0;
    unsigned long long occ[7] = {0, 0, 0, 0, 0, 0, 0};
    T4Lane L; L.mode = T8_IDLE; L.cur = 0; L.lp = 0; L.finish = false;
    WaveQueue Q; wq_init(Q, count, chunk_max);
    uint32_t rid = 0;
    RaySetup R; R.o = V3(0.0f, 0.0f, 0.0f); R.inv = R.o; R.dorig = R.o; R.t_max = 0.0f; R.neg24 = 0; R.kz = 0; R.sx = R.sy = R.sz = 0.0f;
    bool found = false;
    const float4 rlo = make_float4(S.root_lo[0], S.root_lo[1], S.root_lo[2], 0.0f), rhi = make_float4(S.root_hi[0], S.root_hi[1], S.root_hi[2], 0.0f);
    for (;;) {
        const unsigned long long idle = __ballot(L.mode == T8_IDLE);
        if (!Q.exhausted && (uint32_t)__popcll(idle) >= refill) {
            const uint32_t need = (uint32_t)__popcll(idle);
            (void)wq_refill(Q, count, head, lane);
            const uint32_t avail = Q.chunk_end - Q.chunk_next;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            bool hand_back = false; uint32_t q_entry = 0;
            if (COUNT) { occ[5]++; occ[6] += need < avail ? need : avail; }
            if (L.mode == T8_IDLE && rank < avail) {
                float4 a, b;
                q_entry = queue[Q.chunk_next + rank];
                load_queued_ray<true>(W, q_entry, &a, &b, &rid);
                ray_setup<false>(R, a, b);
                St.sp = St.base; L.cur = 0; found = false;
                if (ray_is_exceptional(R.o.x, R.o.y, R.o.z, R.inv.x, R.inv.y, R.inv.z) || ray_out_of_range8(R.o.x, R.o.y, R.o.z, R.inv.x, R.inv.y, R.inv.z)) hand_back = true;
                else if (!slab_test(rlo, rhi, R.o, R.inv, R.t_max)) store_occluded(W, rid, false);       /* the root's own box (bvh.rs:228 at node 0) */
                else L.mode = T8_NODE;
            }
            if (__ballot(hand_back) != 0) wave_push(hand_back, q_entry, W.q_exc_any, &W.counters[CTR(33)]);
            Q.chunk_next += (need < avail ? need : avail);
        }
        const unsigned long long m_node = __ballot(L.mode == T8_NODE), m_leaf = __ballot(L.mode >= T8_LEAF);
        if ((m_node | m_leaf) == 0) { if (Q.exhausted) break; else continue; }
        L.finish = false;
        if (COUNT) occ[0]++;
        if (m_node != 0 && (uint32_t)__popcll(m_leaf) < leaf_batch) {
#pragma unroll
            for (int burst = 0; burst < BURST; burst++) {
                const bool on = L.mode == T8_NODE && !L.finish;
                if (COUNT && on) n_rec++;
                if (COUNT) { const unsigned long long m_on = __ballot(on); if (m_on) { occ[1]++; occ[2] += (unsigned long long)__popcll(m_on); } }
                if (__builtin_expect(__ballot(on && (st_tiny || St.sp > st_soft)) == 0, 1)) { if (on) t8_any_step<false, POLICY>(S, St, L, R.o, R.inv, R.t_max); }
                else if (on) t8_any_step<true, POLICY>(S, St, L, R.o, R.inv, R.t_max);
            }
        } else {
            /* ---- leaf step: the leaf's exact box (on entering the leaf), then one primitive */
            if (COUNT) { occ[3]++; occ[4] += (unsigned long long)__popcll(m_leaf); }
            if (L.mode >= T8_LEAF) {
                const bool entering = L.mode == T8_LEAF;
                uint32_t prim = L.lp & 0x3fffffffu;
                float4 x0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), x1 = x0;
                const bool explicit_box = entering && (L.lp & T8_XBOX);
                if (explicit_box) { x0 = S.oct_xbox[2 * (size_t)prim]; x1 = S.oct_xbox[2 * (size_t)prim + 1]; prim = __float_as_uint(x0.w); }
                float4 g0 = S.geom[S.geom_stride * prim], g1 = S.geom[S.geom_stride * prim + 1], g2 = S.geom[S.geom_stride * prim + 2];
                pin4(g0); pin4(g1); pin4(g2);
                bool pass = true;
                if (entering) {
                    if (!explicit_box) {                               /* Triangle::world_bound (triangle.rs:152-158): the union of the three vertices */
                        x0 = make_float4(fmin_(fmin_(g0.x, g1.x), g2.x), fmin_(fmin_(g0.y, g1.y), g2.y), fmin_(fmin_(g0.z, g1.z), g2.z), 0.0f);
                        x1 = make_float4(fmax_(fmax_(g0.x, g1.x), g2.x), fmax_(fmax_(g0.y, g1.y), g2.y), fmax_(fmax_(g0.z, g1.z), g2.z), 0.0f);
                    }
                    pass = slab_test(x0, x1, R.o, R.inv, R.t_max);     /* the reference's test of the leaf node (bvh.rs:228) */
                }
                bool hh = false;
                if (pass) {
                    if (COUNT) n_prim++;
                    float t = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
                    hh = prim_hit<false>(S, prim, g0, g1, g2, R.o, R.dorig, R.t_max, R.kz, R.sx, R.sy, R.sz, &t, &b0, &b1, &b2);
                }
                if (hh) { found = true; L.finish = true; }
                else if (!pass || (__float_as_uint(g0.w) & GF_LEAF_END)) {
                    if (St.empty()) L.finish = true;
                    else { const uint32_t next = St.template pop<true>(); if (next >> 31) { L.lp = next & 0x7fffffffu; L.mode = T8_LEAF; } else { L.cur = next; L.mode = T8_NODE; } }
                } else { L.lp = prim + 1u; L.mode = T8_LEAF_IN; }
            }
        }
        if (L.finish) { store_occluded(W, rid, found); L.mode = T8_IDLE; }
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { n_rec += __shfl_down(n_rec, off, 64); n_prim += __shfl_down(n_prim, off, 64); }
        if (lane == 0) {
            if (n_rec) { atomicAdd(&stats->quad_records, n_rec); atomicAdd(&stats->quad_records_any, n_rec); }
            if (n_prim) { atomicAdd(&stats->prims_tested, n_prim); atomicAdd(&stats->prims_any, n_prim); }
            for (int k = 0; k < 7; k++) if (occ[k]) atomicAdd(&stats->t4_occ[7 + k], occ[k]);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats->rays_any, (unsigned long long)count);
}

/* ------------------------------------------------------------------ launch
 * LDS per workgroup decides the occupancy: the closest-hit kernel keeps `lds_entries` 8-byte levels per lane in LDS (2 KB per level
 * and workgroup), the any-hit kernel 4-byte levels; deeper levels live in `spill` (per lane, level-major). */
Trace4Plan trace4_plan(const DScene& S, int n_cu, uint32_t knob_entries_closest, uint32_t knob_entries_any, uint32_t knob_wg_closest, uint32_t knob_wg_any, uint32_t knob_wg_oct, uint32_t knob_entries_oct) {
    Trace4Plan p; memset(&p, 0, sizeof(p));
    const uint32_t bound = S.quad_stack_bound ? S.quad_stack_bound : 1u;
    /* workgroups per CU the LDS stacks should leave room for.  Closest-hit: what its registers allow (96 VGPRs -> 5 waves per SIMD -> 5
     * workgroups of 4 waves; 4: +6 % time, 3: +23 %).  Any-hit: 4 (71 VGPRs would allow 7, but 4 measured best: 109.3 ms per step against
     * 112.4 at 6 and 116.5 at 3 -- fewer rays in flight walk the shared upper levels closer together; profiles/r02/e_probe_wg_low.log).
     * 2 KB of the 160 KB stay free: five workgroups of exactly 32 KB were observed to run as four. */
    const uint32_t wg_c = knob_wg_closest ? knob_wg_closest : 5u, wg_a = knob_wg_any ? knob_wg_any : 4u;
    const uint32_t budget_c = ((158u * 1024u) / wg_c) & ~1023u, budget_a = ((158u * 1024u) / wg_a) & ~1023u;
    uint32_t ec = knob_entries_closest ? knob_entries_closest : budget_c / (256u * 8u);
    uint32_t ea = knob_entries_any ? knob_entries_any : budget_a / (256u * 4u);
    p.entries_closest = ec < bound ? ec : bound; p.entries_any = ea < bound ? ea : bound;
    if (p.entries_closest == 0) p.entries_closest = 1;
    if (p.entries_any == 0) p.entries_any = 1;
    p.spill_closest = bound - p.entries_closest; p.spill_any = bound - p.entries_any;
    p.lds_closest = (size_t)p.entries_closest * 256u * 8u; p.lds_any = (size_t)p.entries_any * 256u * 4u;
    auto per_cu = [](size_t lds, uint32_t wg) { return (unsigned)std::max<size_t>(1, std::min<size_t>(wg, (size_t)(158 * 1024) / std::max<size_t>(lds, 1))); };
    p.grid_closest = (unsigned)n_cu * per_cu(p.lds_closest, wg_c); p.grid_any = (unsigned)n_cu * per_cu(p.lds_any, wg_a);
    /* eight-box occlusion records (k_wf_trace8_any): 4-byte entries, their own stack bound */
    if (S.oct && S.n_octs) {
        const uint32_t wg_8 = knob_wg_oct ? knob_wg_oct : 5u, budget_8 = ((158u * 1024u) / wg_8) & ~1023u, bound8 = S.oct_stack_bound ? S.oct_stack_bound : 1u;
        const uint32_t e8 = knob_entries_oct ? knob_entries_oct : budget_8 / (256u * 4u);
        p.entries_oct = e8 < bound8 ? e8 : bound8;
        p.spill_oct = bound8 - p.entries_oct;
        p.lds_oct = (size_t)p.entries_oct * 256u * 4u;
        p.grid_oct = (unsigned)n_cu * per_cu(p.lds_oct, wg_8);
        p.oct_ok = true;
    }
    return p;
}

void launch_trace4(int kind, bool count, bool spheres, unsigned grid, size_t lds, uint32_t lds_entries, void* spill, hipStream_t stream, const DScene& S, const WfBuffers& W,
                   const uint32_t* queue, const uint32_t* count_ptr, uint32_t* head, DevStats* stats, uint32_t refill, uint32_t leaf_batch, uint32_t chunk, uint32_t burst, uint32_t any_policy, uint32_t spill_levels) {
    const bool any = kind != T4K_CLOSEST;
#define FTN_T4(K, SPILL_T, ...) hipLaunchKernelGGL((K<__VA_ARGS__>), dim3(grid), dim3(256), lds, stream, S, W, queue, count_ptr, head, stats, refill, leaf_batch, chunk, lds_entries, (SPILL_T)spill, spill_levels)
    if (kind == T4K_ANY_OCT) {        /* (triangle-only scenes with eight-box records: the caller checked) */
#define FTN_T8(C, B) do { if (any_policy == 0) FTN_T4(k_wf_trace8_any, uint32_t*, C, B, 0); else FTN_T4(k_wf_trace8_any, uint32_t*, C, B, 1); } while (0)
        if (count) FTN_T8(true, 2);
        else if (burst <= 1) FTN_T8(false, 1); else if (burst == 2) FTN_T8(false, 2); else FTN_T8(false, 3);
#undef FTN_T8
    } else if (any) {
#define FTN_T4A(C, Sp, B) do { if (any_policy == 0) FTN_T4(k_wf_trace4_any, uint32_t*, C, Sp, B, 0); else FTN_T4(k_wf_trace4_any, uint32_t*, C, Sp, B, 1); } while (0)
        if (count) { if (spheres) FTN_T4A(true, true, 2); else FTN_T4A(true, false, 2); }
        else if (spheres) { if (burst <= 2) FTN_T4A(false, true, 2); else FTN_T4A(false, true, 4); }
        else { if (burst <= 1) FTN_T4A(false, false, 1); else if (burst == 2) FTN_T4A(false, false, 2); else if (burst == 3) FTN_T4A(false, false, 3); else FTN_T4A(false, false, 4); }
#undef FTN_T4A
    } else {
        if (count) { if (spheres) FTN_T4(k_wf_trace4, uint2*, true, true, 2); else FTN_T4(k_wf_trace4, uint2*, true, false, 2); }
        else if (spheres) { if (burst <= 2) FTN_T4(k_wf_trace4, uint2*, false, true, 2); else FTN_T4(k_wf_trace4, uint2*, false, true, 4); }
        else { if (burst <= 1) FTN_T4(k_wf_trace4, uint2*, false, false, 1); else if (burst == 2) FTN_T4(k_wf_trace4, uint2*, false, false, 2); else if (burst == 3) FTN_T4(k_wf_trace4, uint2*, false, false, 3); else FTN_T4(k_wf_trace4, uint2*, false, false, 4); }
    }
#undef FTN_T4
}

}  // namespace ftn
