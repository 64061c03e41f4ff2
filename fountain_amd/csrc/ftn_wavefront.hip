/*
 * ftn_wavefront.hip -- wavefront restructuring of PathIntegrator::incident_radiance (src/integrator/path.rs:25-95) for gfx950.
 *
 * The reference walks one path at a time: intersect -> emission -> NEE (shadow ray + MIS ray) -> BSDF sample -> RR -> loop.
 * Here a PASS renders S samples of every owned pixel at once (one wavefront of up to 256 Mi paths, path id = slot * S + sample);
 * paths live as float4 SoA records in HBM and every bounce runs these kernels over queues of path / ray ids:
 *
 *   k_wf_generate    get_camera_sample + generate_ray (sampler/mod.rs:43-51, camera/mod.rs:145-205); all paths -> closest queue
 *   k_wf_trace<ANY>  Scene::intersect / intersect_test (bvh.rs:160-266) for a queue of rays.  Persistent 256-thread
 *                    workgroups; each wave64 pulls rays from the queue in chunks (one atomic per chunk) and re-arms idle lanes
 *                    by __ballot / __popcll ranks: lanes whose ray finished get a new one while the others keep walking.
 *                    Per-lane node stack in LDS, [level][lane] interleaved.  The queue is cut into one slice per XCD.
 *   k_wf_trace_any2  the production any-hit walk (shadow rays, MIS rays toward infinite lights): two boxes per step from 64-byte
 *                    records, children in the order that ends blocked rays soonest -- a boolean does not depend on the order
 *   k_wf_classify    groups the active paths by shading class (finished / missed / one per material type / null material)
 *   k_wf_shade<TEX, MT, ENV>  everything between two intersect calls of the Li loop: resolves the previous bounce's direct lighting
 *                    (shadow + MIS results), emission, termination, Material -> Bsdf, uniform_sample_one_light /
 *                    estimate_direct set-up, BSDF sampling, Russian roulette; emits shadow / MIS / continuation rays and
 *                    compacts the surviving paths into the next queue (one atomic per workgroup and queue).  One launch per
 *                    material type present, BSDF code specialised at compile time; ENV: the scene's only light is an
 *                    InfiniteAreaLight whose record travels in the kernel arguments.
 *   k_wf_ray_keys + rocPRIM radix sort: orders the closest-hit queue by Morton(origin cell) | direction octant (coherence)
 *   k_wf_accumulate  Film::add_sample_to_tile for the pass's samples of each pixel IN SAMPLE ORDER (film.rs:136-172).
 *
 * Small wavefronts (<= 32 Mi paths) run the any-hit trace of a bounce on a second stream beside the closest-hit trace.
 *
 * All per-path arithmetic and the order of the RNG draws are those of the reference (and of the megakernel), so both
 * pipelines produce identical radiance.  Dominant kernel: k_wf_trace<false>; roofline = HBM (node + triangle fetches), in
 * practice the L2 -> L1 gather rate.
 */
#include "ftn_wf_common.h"
#include "ftn_texture.h"
#include <string>
#include <cstdlib>
#include <hipcub/hipcub.hpp>

namespace ftn {

/* pstate bits */
/* ------------------------------------------------------------------ generate (no atomics: queue slots are known in closed form) */
__global__ void __launch_bounds__(256) k_wf_generate(RenderParams P, WfBuffers W, int write_state) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;          /* path id = slot * samples + s: the samples of a pixel are neighbours */
    if (i >= W.n_paths) return;
    const uint32_t slot = i / W.samples, s = i % W.samples;
    const DTile tile = P.tiles[slot >> 8];
    const int px = tile.x0 + (int)(slot & 15u), py = tile.y0 + (int)((slot >> 4) & 15u);
    if (px < tile.x1 && py < tile.y1) {
        Rng rng; rng.seed(indexed_key(P.seed, px, py, W.first_sample + s));
        V2 j = rng.next2();
        V2 p_film((float)px + j.x, (float)py + j.y);
        V2 p_lens = rng.next2();
        float time_u = rng.next();
        DRay ray = camera_ray(P.C, p_film, p_lens, time_u);
        W.ray[2 * (size_t)(i)] = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.0f);
        W.ray[2 * (size_t)(i) + 1] = make_float4(ray.d.x, ray.d.y, ray.d.z, ray.t_max);
        /* A fresh path's state is a constant (throughput 1, radiance 0) and a stream position that follows from the sample's key: the path
         * integrator's first shading pass rebuilds it (k_wf_shade, `first`) instead of reading 64 bytes per path that this kernel would
         * have to write.  The direct-lighting / Whitted stage still reads it. */
        if (write_state) {
            W.beta[i] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(PS_ALIVE));
            W.rad[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            W.rng01[i] = make_ulonglong2(rng.s0, rng.s1); W.rng23[i] = make_ulonglong2(rng.s2, rng.s3);
        }
        /* (the film position is not kept: k_wf_accumulate draws it again from the sample's key) */
        /* queue order = path order: tile, pixel, sample.  A wave of the first trace (and of the first shading pass, and of the shadow
         * rays it emits) covers a few pixels instead of 64, each XCD's slice of the queue is a part of the image rather than one whole
         * sample of it, and the path state is still read in order. */
        uint32_t local = (uint32_t)(py - tile.y0) * (uint32_t)(tile.x1 - tile.x0) + (uint32_t)(px - tile.x0);
        if (W.gen_blocks && tile.x1 - tile.x0 == 16 && tile.y1 - tile.y0 == 16) {          /* Morton order inside a full tile: x bits at even, y bits at odd positions */
            const uint32_t x = slot & 15u, y = (slot >> 4) & 15u;
            local = (x & 1u) | ((y & 1u) << 1) | ((x & 2u) << 1) | ((y & 2u) << 2) | ((x & 4u) << 2) | ((y & 4u) << 3) | ((x & 8u) << 3) | ((y & 8u) << 4);
        }
        const uint32_t pix = tile.valid_off + local;
        const uint32_t qi = pix * W.samples + s;
        W.q_active[0][qi] = i | (P.max_depth == 0 ? WF_Q_DEPTH : 0u);
        W.q_closest[qi] = i;
    }
}

/* ------------------------------------------------------------------ trace
 * Per lane: mode 0 = needs a ray, 1 = at a BVH node, 2 = holds a leaf whose primitives are still to be tested.
 * A wave alternates between two cheap, convergent bodies instead of running the (long) triangle test for the two or three
 * lanes that happen to sit on a leaf at every step:
 *     node step   for all lanes in mode 1                         (~40 VALU instructions)
 *     leaf step   for all lanes in mode 2, once >= leaf_batch lanes wait there or no lane is left in mode 1
 * The visiting order of every single ray is exactly the reference's (bvh.rs:160-266): near child first, the far child is
 * re-tested against the shrunken t_max when popped, leaf primitives in order, `t == t_max` accepted.
 * Rays are taken from the queue in per-wave chunks (one global atomic per `chunk` rays) and lanes are re-armed from the chunk
 * with __ballot / __popcll prefix ranks as soon as `refill` of them are idle. */
enum : uint32_t { TM_IDLE = 0, TM_NODE = 1, TM_LEAF = 2 };

template <bool ANY, bool COUNT, bool SPHERES, int BURST /* 0: node_burst is a runtime value */>
__global__ void __launch_bounds__(256) k_wf_trace(DScene S, WfBuffers W, const uint32_t* __restrict__ queue, const uint32_t* count_ptr, uint32_t* head,
                                                  DevStats* stats, uint32_t refill, uint32_t leaf_batch, uint32_t chunk, uint32_t node_burst) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t* const st_base = lds_stack + threadIdx.x;      /* [level][lane]; the stack pointer is kept as an LDS address */
    uint32_t* sptr = st_base;
    const uint32_t count = *count_ptr;
    const uint32_t lane = lane_id();
    const uint32_t np = W.n_paths;
    TravCount tc{0, 0};
    uint32_t w_idle_lanes = 0, w_leafwait_lanes = 0;
    uint32_t w_node_steps = 0, w_leaf_steps = 0;   /* COUNT only: wave-level step counts -> SIMD lane utilisation (FTN_WF_DEBUG) */
    uint32_t mode = TM_IDLE;
    uint32_t chunk_next = 0, chunk_end = 0; bool exhausted = count == 0;
    /* The queue is cut into 8 contiguous slices, one per XCD (workgroups b and b + 8 share an XCD and therefore an L2): with the
     * queue sorted by origin cell, an XCD's L2 only has to hold the part of the BVH its slice of space reaches.  A wave whose slice
     * has run dry moves on to the next one, so the partition costs no load balance. */
    uint32_t slice = blockIdx.x & 7u, slices_done = 0;
    /* per-wave chunk: few queue-head atomics, but small enough that the tail spreads over all waves */
    { uint32_t c = count / (gridDim.x * 4u * 16u); c &= ~63u; chunk = c < 64u ? 64u : (c > chunk ? chunk : c); }
    uint32_t rid = 0, cur = 0 /* byte offset of the node record */, neg16 = 0, lp = 0, lp_end = 0;
    V3 o, inv, dperm; float t_max = 0.0f, sx = 0.0f, sy = 0.0f, sz = 0.0f; int kz = 0;
    int hprim = -1; float hb0 = 0.0f, hb1 = 0.0f, hb2 = 0.0f; bool found = false;
    V3 dorig;   /* only the sphere path needs the unpermuted direction */
    for (;;) {
        /* ---- re-arm idle lanes */
        const unsigned long long idle = __ballot(mode == TM_IDLE);
        if (!exhausted && (uint32_t)__popcll(idle) >= refill) {
            const uint32_t need = (uint32_t)__popcll(idle);
            while (chunk_next == chunk_end && !exhausted) {      /* the wave's chunk is used up: take a new one from the current slice */
                const uint32_t s_lo = (uint32_t)(((unsigned long long)count * slice) >> 3) & ~63u, s_hi = slice == 7u ? count : ((uint32_t)(((unsigned long long)count * (slice + 1u)) >> 3) & ~63u);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(head + CTR(slice), chunk);
                base = __shfl(base, 0, 64) + s_lo;
                if (base >= s_hi) {                                       /* this slice is empty: help the next XCD's */
                    slice = (slice + 1u) & 7u;
                    if (++slices_done == 8u) {
                        exhausted = true;
                        /* the launch starts to drain: let the stream that waits for it (the any-hit trace) go ahead */
                        if (!ANY && W.drain_sig && lane == 0 && atomicAdd(&W.counters[CTR(11)], 1u) + 1u == W.drain_at)      /* the drain_at-th wave of this launch */
                            __hip_atomic_store(W.drain_sig, W.drain_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);          /* a timing hint only: no data is handed over */
                    }
                }
                else { chunk_next = base; chunk_end = base + chunk < s_hi ? base + chunk : s_hi; }
            }
            const uint32_t avail = chunk_end - chunk_next;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (mode == TM_IDLE && rank < avail) {
                float4 a, b;
                load_queued_ray<ANY>(W, queue[chunk_next + rank], &a, &b, &rid);      /* rid: the result slot of this ray from here on */
                o = V3(a.x, a.y, a.z); const V3 d(b.x, b.y, b.z); t_max = b.w;
                if (SPHERES) dorig = d;
                inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                neg16 = (d.x < 0.0f ? 0x10000u : 0u) | (d.y < 0.0f ? 0x20000u : 0u) | (d.z < 0.0f ? 0x40000u : 0u);   /* dir_is_neg, aligned with the node's one-hot axis */
                /* Triangle::intersect's per-ray constants (triangle.rs:189-205): permutation and shear depend on the ray only */
                kz = max_dimension(vabs(d));
                { const int kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1; dperm = V3(d.get(kx), d.get(ky), d.get(kz)); }
                sx = -dperm.x / dperm.z; sy = -dperm.y / dperm.z; sz = 1.0f / dperm.z;
                sptr = st_base; cur = 0; found = false; hprim = -1; hb0 = 0.0f; hb1 = 0.0f; hb2 = 0.0f;
                mode = TM_NODE;
                if (S.n_nodes == 0) {   /* empty scene: immediate miss */
                    if (ANY) store_occluded(W, rid, false); else { W.hit[rid] = make_float4(FTN_INF, 0.0f, 0.0f, 0.0f); W.hit_prim[rid] = -1; }
                    mode = TM_IDLE;
                }
            }
            chunk_next += (need < avail ? need : avail);
        }
        const unsigned long long m_node = __ballot(mode == TM_NODE), m_leaf = __ballot(mode == TM_LEAF);
        if ((m_node | m_leaf) == 0) { if (exhausted) break; else continue; }
        bool finish = false;
        if (m_node != 0 && (uint32_t)__popcll(m_leaf) < leaf_batch) {
            /* ---- node steps: `node_burst` of them per control round (lanes that reach a leaf or finish sit out the rest) */
#pragma unroll
            for (uint32_t burst = 0; burst < (BURST ? (uint32_t)BURST : node_burst); burst++) {
                if (COUNT && __ballot(mode == TM_NODE && !finish) != 0) { w_node_steps++; w_idle_lanes += (uint32_t)__popcll(__ballot(mode == TM_IDLE || finish)); w_leafwait_lanes += (uint32_t)__popcll(__ballot(mode == TM_LEAF)); }
                if (mode == TM_NODE && !finish) {
                    const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.nodes) + cur);
                    float4 na = rec[0], nb = rec[1];
                    pin4(na); pin4(nb);
                    if (COUNT) tc.nodes++;
                    bool pop = true;
                    if (slab_test_node(na, nb, o, inv, t_max)) {
                        const uint32_t idx = __float_as_uint(nb.z), meta = __float_as_uint(nb.w);
                        if (meta >> 24) { lp = idx; lp_end = idx + (meta & 0xffffu); mode = TM_LEAF; pop = false; }
                        else {
                            if (meta & neg16) { *sptr = cur + 32u; cur = idx; }      /* dir_is_neg[axis]: second child first */
                            else { *sptr = idx; cur = cur + 32u; }
                            sptr += 256;
                            pop = false;
                        }
                    }
                    if (pop) { if (sptr == st_base) finish = true; else { sptr -= 256; cur = *sptr; } }
                }
            }
        } else {
            /* ---- leaf step: one primitive per lane */
            if (COUNT) w_leaf_steps++;
            if (mode == TM_LEAF) {
                const uint32_t prim = lp;
                float4 g0 = S.geom[S.geom_stride * prim], g1 = S.geom[S.geom_stride * prim + 1], g2 = S.geom[S.geom_stride * prim + 2];
                pin4(g0); pin4(g1); pin4(g2);
                if (COUNT) tc.prims++;
                float t = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
                const bool hh = prim_hit<SPHERES>(S, prim, g0, g1, g2, o, dorig, t_max, kz, sx, sy, sz, &t, &b0, &b1, &b2);
                if (hh) { found = true; t_max = t; hprim = (int)prim; hb0 = b0; hb1 = b1; hb2 = b2; }
                lp++;
                if (ANY && hh) finish = true;
                else if (lp == lp_end) { if (sptr == st_base) finish = true; else { sptr -= 256; cur = *sptr; mode = TM_NODE; } }
            }
        }
        if (finish) {
            if (ANY) store_occluded(W, rid, found);
            else { W.hit[rid] = make_float4(found ? t_max : FTN_INF, hb0, hb1, hb2); W.hit_prim[rid] = hprim; }
            mode = TM_IDLE;
        }
    }
    if (COUNT) {
        unsigned long long n = tc.nodes, p = tc.prims;
        for (int off = 32; off > 0; off >>= 1) { n += __shfl_down(n, off, 64); p += __shfl_down(p, off, 64); }
        if (lane == 0) {
            if (n) atomicAdd(&stats->nodes_visited, n); if (p) atomicAdd(&stats->prims_tested, p);
            if (ANY) { if (n) atomicAdd(&stats->nodes_any, n); if (p) atomicAdd(&stats->prims_any, p); }
            atomicAdd(&W.counters[CTR(6)], w_node_steps); atomicAdd(&W.counters[CTR(7)], w_leaf_steps);
            atomicAdd(&W.counters[CTR(8)], w_idle_lanes >> 6); atomicAdd(&W.counters[CTR(9)], w_leafwait_lanes >> 6);     /* in units of 64 lanes */
        }
    }
    if (stats && blockIdx.x == 0 && threadIdx.x == 0) { if (ANY) atomicAdd(&stats->rays_any, (unsigned long long)count); else atomicAdd(&stats->rays_closest, (unsigned long long)count); }   /* (NULL: the rays were counted by the kernel that handed them over) */
}

/* ------------------------------------------------------------------ any-hit traversal over two-box records
 *
 * Scene::intersect_test returns a boolean, and t_max never shrinks while it walks: whether a box or a triangle is reached does not
 * depend on the order of the walk.  So the any-hit kernel is free to test BOTH children of an interior node in one step from one
 * 64-byte record (DScene::fat: the two children's node records side by side) and to descend only into children whose box the ray
 * enters -- a node that fails its test is never fetched, and a step advances two levels' worth of box tests per memory round trip.
 * (For closest hits the far child must be re-tested against the shrunken t_max when it is popped; carrying its entry distance on
 * the stack doubles the LDS per lane and was measured slower, see DESIGN.md.)  The stack entry is one word: byte offset of an
 * interior child's record, or first primitive | bit 31 for a leaf child; leaf primitives are walked until GF_LEAF_END.
 * The counting build keeps k_wf_trace<ANY> (the reference's walk) so that node tallies equal the oracle's. */
template <bool SPHERES>
__global__ void __launch_bounds__(256) k_wf_trace_any2(DScene S, WfBuffers W, const uint32_t* __restrict__ queue, const uint32_t* count_ptr, uint32_t* head,
                                                       DevStats* stats, uint32_t refill, uint32_t leaf_batch, uint32_t chunk, uint32_t policy) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t* const st_base = lds_stack + threadIdx.x;
    uint32_t* sptr = st_base;
    const uint32_t count = *count_ptr;
    const uint32_t lane = lane_id();
    uint32_t mode = TM_IDLE;
    uint32_t chunk_next = 0, chunk_end = 0; bool exhausted = count == 0;
    uint32_t slice = blockIdx.x & 7u, slices_done = 0;                     /* one queue slice per XCD, see k_wf_trace */
#ifdef FTN_DRAIN_PROBE      /* experiment build (tools/gpu_drain_probe.py): when does the first / last wave find the queue dry, when does the launch end */
    const uint32_t t_start = (uint32_t)wall_clock64(); uint32_t t_dry = t_start; bool seen_dry = false;
#endif
    { uint32_t c = count / (gridDim.x * 4u * 16u); c &= ~63u; chunk = c < 64u ? 64u : (c > chunk ? chunk : c); }
    uint32_t rid = 0, cur = 0, neg16 = 0, lp = 0;
    V3 o, inv, dperm; float t_max = 0.0f, sx = 0.0f, sy = 0.0f, sz = 0.0f; int kz = 0;
    bool found = false;
    V3 dorig;
    const float4 rlo = make_float4(S.root_lo[0], S.root_lo[1], S.root_lo[2], 0.0f), rhi = make_float4(S.root_hi[0], S.root_hi[1], S.root_hi[2], 0.0f);
    for (;;) {
        const unsigned long long idle = __ballot(mode == TM_IDLE);
        if (!exhausted && (uint32_t)__popcll(idle) >= refill) {
            const uint32_t need = (uint32_t)__popcll(idle);
            while (chunk_next == chunk_end && !exhausted) {
                const uint32_t s_lo = (uint32_t)(((unsigned long long)count * slice) >> 3) & ~63u, s_hi = slice == 7u ? count : ((uint32_t)(((unsigned long long)count * (slice + 1u)) >> 3) & ~63u);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(head + CTR(slice), chunk);
                base = __shfl(base, 0, 64) + s_lo;
                if (base >= s_hi) { slice = (slice + 1u) & 7u; if (++slices_done == 8u) exhausted = true; }
                else { chunk_next = base; chunk_end = base + chunk < s_hi ? base + chunk : s_hi; }
            }
            const uint32_t avail = chunk_end - chunk_next;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (mode == TM_IDLE && rank < avail) {
                float4 a, b;
                load_queued_ray<true>(W, queue[chunk_next + rank], &a, &b, &rid);
                o = V3(a.x, a.y, a.z); const V3 d(b.x, b.y, b.z); t_max = b.w;
                if (SPHERES) dorig = d;
                inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                neg16 = (d.x < 0.0f ? 0x10000u : 0u) | (d.y < 0.0f ? 0x20000u : 0u) | (d.z < 0.0f ? 0x40000u : 0u);
                kz = max_dimension(vabs(d));
                { const int kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1; dperm = V3(d.get(kx), d.get(ky), d.get(kz)); }
                sx = -dperm.x / dperm.z; sy = -dperm.y / dperm.z; sz = 1.0f / dperm.z;
                sptr = st_base; cur = 0; found = false;
                /* the root's own box test (bvh.rs:228-230) */
                if (S.n_nodes == 0 || !slab_test(rlo, rhi, o, inv, t_max)) { store_occluded(W, rid, false); mode = TM_IDLE; }
                else if (S.root_is_leaf) { lp = 0; mode = TM_LEAF; }
                else mode = TM_NODE;
            }
            chunk_next += (need < avail ? need : avail);
        }
        const unsigned long long m_node = __ballot(mode == TM_NODE), m_leaf = __ballot(mode == TM_LEAF);
#ifdef FTN_DRAIN_PROBE
        if (exhausted && !seen_dry) { seen_dry = true; t_dry = (uint32_t)wall_clock64(); }
#endif
        if ((m_node | m_leaf) == 0) { if (exhausted) break; else continue; }
        bool finish = false;
        if (m_node != 0 && (uint32_t)__popcll(m_leaf) < leaf_batch) {
#pragma unroll
            for (uint32_t burst = 0; burst < 8u; burst++) {
                if (mode == TM_NODE && !finish) {
                    const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.fat) + cur);
                    float4 a0 = rec[0], b0 = rec[1], a1 = rec[2], b1 = rec[3];
                    pin4(a0); pin4(b0); pin4(a1); pin4(b1);
                    float i00, i01, i10, i11;
                    const bool h0 = slab_test_node_iv(a0, b0, o, inv, t_max, &i00, &i01), h1 = slab_test_node_iv(a1, b1, o, inv, t_max, &i10, &i11);
                    const uint32_t m0 = __float_as_uint(b0.w), m1 = __float_as_uint(b1.w);
                    /* stack / next entries: byte offset of the child's record, or first primitive | bit 31 for a leaf child */
                    const uint32_t e0 = __float_as_uint(b0.z) | ((m0 >> 24) ? 0x80000000u : 0u), e1 = __float_as_uint(b1.z) | ((m1 >> 24) ? 0x80000000u : 0u);
                    /* Which child first is a free choice here (the boolean does not depend on it) and it matters: a ray that is blocked at
                     * all ends at its first hit.  Measured on the config-5 scene, per step: reference order (near side of the split plane
                     * first, policy 0) 34.3 ms; far side first (1) 32.1 ms -- shadow and MIS rays start on a surface, what blocks them
                     * tends to lie ahead rather than around the origin; the child whose box the ray stays in longest first (2) 31.9 ms;
                     * leaf-child-first, interior-child-first, shortest stay, larger entry / exit distance first: slower or equal. */
                    const bool second_first = (policy & 0xffu) == 2u ? (i11 - i10) > (i01 - i00) : (((m0 & neg16) != 0) != ((policy & 0xffu) == 1u));
                    const uint32_t en = second_first ? e1 : e0, ef = second_first ? e0 : e1;
                    const bool hn = second_first ? h1 : h0, hf = second_first ? h0 : h1;
                    uint32_t next = 0; bool have = true;
                    if (hn) { next = en; if (hf) { *sptr = ef; sptr += 256; } }
                    else if (hf) next = ef;
                    else if (sptr == st_base) { finish = true; have = false; }
                    else { sptr -= 256; next = *sptr; }
                    if (have) { if (next >> 31) { lp = next & 0x7fffffffu; mode = TM_LEAF; } else cur = next; }
                }
            }
        } else {
            if (mode == TM_LEAF) {
                const uint32_t prim = lp;
                float4 g0 = S.geom[S.geom_stride * prim], g1 = S.geom[S.geom_stride * prim + 1], g2 = S.geom[S.geom_stride * prim + 2];
                pin4(g0); pin4(g1); pin4(g2);
                float t = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
                const bool hh = prim_hit<SPHERES>(S, prim, g0, g1, g2, o, dorig, t_max, kz, sx, sy, sz, &t, &b0, &b1, &b2);
                if (hh) { found = true; finish = true; }
                else if (__float_as_uint(g0.w) & GF_LEAF_END) {
                    if (sptr == st_base) finish = true;
                    else { sptr -= 256; const uint32_t next = *sptr; if (next >> 31) lp = next & 0x7fffffffu; else { cur = next; mode = TM_NODE; } }
                } else lp++;
            }
        }
        if (finish) { store_occluded(W, rid, found); mode = TM_IDLE; }
    }
#ifdef FTN_DRAIN_PROBE
    if (lane == 0) {
        uint32_t* T = &W.counters[CTR(13) + 4 * ((policy >> 8) & 7u)];             /* one slot of four words per bounce */
        atomicMax(&T[0], ~t_start); atomicMax(&T[1], ~t_dry); atomicMax(&T[2], t_dry); atomicMax(&T[3], (uint32_t)wall_clock64());
    }
#endif
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats->rays_any, (unsigned long long)count);
}

/* ------------------------------------------------------------------ material-sorted shading: group the active queue by shading class
 * class 0: finished paths that only need their pending direct term resolved; 1: alive, ray escaped or depth limit reached
 * (emission / environment only); 2..: alive hit, by material type (2 + FTN_MAT_*), 7: primitive without material.
 * Two passes (count, scatter) with block-level ballots; class segments start at multiples of 256 so that every shade
 * workgroup sees a single class and its branches on hit/material are wave-uniform. */
#define WF_NCLASS 8
__device__ inline uint32_t wf_class_of(const RenderParams& P, const WfBuffers& W, uint32_t entry) {
    /* an active-queue entry says whether the path has ended / reached max_depth (WF_Q_*), hit_prim is a dense array, the primitive's
     * class is a byte (DScene::prim_class): nothing here gathers from the path state or the 32-byte prim_info records */
    if (entry & WF_Q_FIN) return 0u;
    const int prim = W.hit_prim[entry & WF_Q_ID_MASK];
    if (prim < 0 || (entry & WF_Q_DEPTH)) return 1u;
    return P.S.prim_class[prim];
}
/* Each class owns a segment of `seg_cap` slots in q_sorted (worst case: every path in one class).  A workgroup classifies all of
 * its rounds first (class codes and per-wave counts stay in LDS), reserves its slots with ONE global atomic per class -- with one
 * atomic per class and round the kernel ran at the rate of same-address atomics (65 k of them on the busiest counter) -- and
 * then writes its paths.  The shade kernel derives the 256-aligned virtual layout from the 8 counts itself. */
#define WF_CLS_ROUNDS 32
__global__ void __launch_bounds__(256) k_wf_classify(RenderParams P, WfBuffers W, int in_q) {
    __shared__ unsigned char s_cls[WF_CLS_ROUNDS][256];
    __shared__ uint32_t s_cnt[WF_CLS_ROUNDS][WF_NCLASS][4];        /* per round, class, wave */
    __shared__ uint32_t s_pre[WF_CLS_ROUNDS][WF_NCLASS];           /* slots of the class used by earlier rounds of this workgroup */
    __shared__ uint32_t s_base[WF_NCLASS];
    const uint32_t count = W.counters[CTR(in_q == 0 ? 0 : 1)];
    const uint32_t stride = gridDim.x * 256u, rounds = (count + stride - 1) / stride;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    for (uint32_t r0 = 0; r0 < rounds; r0 += WF_CLS_ROUNDS) {
        const uint32_t nr = rounds - r0 < WF_CLS_ROUNDS ? rounds - r0 : WF_CLS_ROUNDS;
        for (uint32_t r = 0; r < nr; r++) {                        /* 1: classify */
            const uint32_t qi = (r0 + r) * stride + blockIdx.x * 256u + threadIdx.x;
            uint32_t c = WF_NCLASS;
            if (qi < count) c = wf_class_of(P, W, W.q_active[in_q][qi]);
            s_cls[r][threadIdx.x] = (unsigned char)c;
#pragma unroll
            for (int k = 0; k < WF_NCLASS; k++) {
                const unsigned long long m = __ballot(c == (uint32_t)k);
                if (lane == 0) s_cnt[r][k][wave] = (uint32_t)__popcll(m);
            }
        }
        __syncthreads();
        if (threadIdx.x < WF_NCLASS) {                             /* 2: reserve */
            const uint32_t k = threadIdx.x; uint32_t tot = 0;
            for (uint32_t r = 0; r < nr; r++) { s_pre[r][k] = tot; tot += s_cnt[r][k][0] + s_cnt[r][k][1] + s_cnt[r][k][2] + s_cnt[r][k][3]; }
            s_base[k] = tot ? atomicAdd(&W.cls[CTR(k)], tot) : 0u;
        }
        __syncthreads();
        for (uint32_t r = 0; r < nr; r++) {                        /* 3: write */
            const uint32_t qi = (r0 + r) * stride + blockIdx.x * 256u + threadIdx.x;
            const uint32_t c = s_cls[r][threadIdx.x];
            unsigned long long mine = 0;
#pragma unroll
            for (int k = 0; k < WF_NCLASS; k++) { const unsigned long long m = __ballot(c == (uint32_t)k); if (c == (uint32_t)k) mine = m; }
            if (c < WF_NCLASS) {
                uint32_t off = s_base[c] + s_pre[r][c];
                for (uint32_t w2 = 0; w2 < wave; w2++) off += s_cnt[r][c][w2];
                W.q_sorted[(size_t)c * W.seg_cap + off + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull))] = W.q_active[in_q][qi] & WF_Q_ID_MASK;
            }
        }
        __syncthreads();
    }
}

/* ------------------------------------------------------------------ shade */
__device__ inline DHit load_hit(const WfBuffers& W, uint32_t r) {
    const float4 h = W.hit[r]; DHit o; o.t = h.x; o.b0 = h.y; o.b1 = h.z; o.b2 = h.w; o.prim = W.hit_prim[r]; return o;
}

#ifndef FTN_SHADE_MIN_WAVES
#define FTN_SHADE_MIN_WAVES 1
#endif
#ifndef FTN_SHADE_WAVES_NOBSDF
#define FTN_SHADE_WAVES_NOBSDF 4      /* finished paths / misses (5 waves: 96 registers, no scratch -- measured equal: 99.3 against 99.5 ms of shading) */
#endif
#ifndef FTN_SHADE_WAVES_MATTE
#define FTN_SHADE_WAVES_MATTE 3
#endif
#ifndef FTN_SHADE_WAVES_OTHER
#define FTN_SHADE_WAVES_OTHER 3
#endif
/* MT: -1 = every class in `class_mask`, material type read at run time (textured scenes);  0..4 = ONE material class, its BSDF code
 * specialised at compile time (fewer registers: 3 waves/SIMD instead of 2, 4 for mirrors);  -2 = the classes without a BSDF
 * (finished paths, misses / depth limit, null materials), the material branch compiled out. */
/* ENV: the scene's only light is an InfiniteAreaLight (DScene::env_only).  Its record is read from the kernel arguments (scalar
 * loads instead of one gather per field and lane) and the point / distant / area-light code is compiled out; every value computed
 * is the one the generic kernel computes for such a scene. */
template <bool TEX, int MT, bool ENV>
__global__ void __launch_bounds__(256, (MT == -1 ? FTN_SHADE_MIN_WAVES : (MT == -2 ? FTN_SHADE_WAVES_NOBSDF : (MT == 0 ? FTN_SHADE_WAVES_MATTE : FTN_SHADE_WAVES_OTHER)))) k_wf_shade(RenderParams P, WfBuffers W, int in_q, uint32_t class_mask, uint32_t first /* 1: the pass right behind k_wf_generate -- every path is fresh */) {
    const DScene& S = P.S;
    /* virtual, 256-aligned concatenation of the selected class segments: a workgroup never straddles two classes */
    uint32_t cbase[WF_NCLASS + 1], ccnt[WF_NCLASS];
    cbase[0] = 0;
#pragma unroll
    for (int k = 0; k < WF_NCLASS; k++) { ccnt[k] = ((class_mask >> k) & 1u) ? W.cls[CTR(k)] : 0u; cbase[k + 1] = cbase[k] + ((ccnt[k] + 255u) & ~255u); }
    const uint32_t count = cbase[WF_NCLASS];
    uint32_t* out_q = W.q_active[in_q ^ 1];
    uint32_t* out_count = &W.counters[CTR(in_q == 0 ? 1 : 0)];
    int err = 0;
    const uint32_t stride = gridDim.x * 256u;
    const uint32_t rounds = (count + stride - 1) / stride;
    for (uint32_t round = 0; round < rounds; round++) {
        const uint32_t qi = round * stride + blockIdx.x * 256u + threadIdx.x;
        bool have = false;
        uint32_t sorted_idx = 0;
        if (qi < count) {                                  /* which class segment does this slot belong to, and is it occupied? */
            uint32_t c = 0, cb = 0, cc = ccnt[0];
#pragma unroll
            for (int k = 1; k < WF_NCLASS; k++) if (qi >= cbase[k]) { c = (uint32_t)k; cb = cbase[k]; cc = ccnt[k]; }
            have = qi - cb < cc;
            sorted_idx = c * W.seg_cap + (qi - cb);
        }
        bool push_active = false, push_closest = false, push_mis = false, push_shadow = false, push_mis_any = false;
        uint32_t p = 0, active_entry = 0;
        if (have) {
            p = W.q_sorted[sorted_idx];
            float4 bq = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(PS_ALIVE)), lq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);       /* a fresh path (k_wf_generate) */
            if (!first) { if (W.br) { bq = W.br[2 * (size_t)p]; lq = W.br[2 * (size_t)p + 1]; } else { bq = W.beta[p]; lq = W.rad[p]; } }
            Rgb beta(bq.x, bq.y, bq.z), L(lq.x, lq.y, lq.z);
            uint32_t ps = __float_as_uint(bq.w);
            /* ---- finish the previous bounce's estimate_direct (integrator/mod.rs:330-392) */
            if (ps & PS_DIRECT) {
                float4 q0, q1, q2, q3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (W.pd) { const float4* Q = W.pd + 4 * (size_t)p; q0 = Q[0]; q1 = Q[1]; q2 = Q[2]; if (W.pd_md) q3 = Q[3]; } else { q0 = W.pend0[p]; q1 = W.pend1[p]; q2 = W.pend2[p]; }
                const uint32_t occ_word = __float_as_uint(q3.w);      /* pd_occ: the any-hit results of this path's two rays (store_occluded) */
                Rgb radiance(0.0f);
                if ((ps & PS_SHADOW) && !(W.pd_occ ? (occ_word & 0xffu) != 0u : W.occluded[p] != 0)) radiance = radiance + Rgb(q0.x, q0.y, q0.z);
                if (ps & PS_MIS_ANY) {                           /* infinite light: the BSDF-sampled ray either escapes to it or contributes nothing (mod.rs:367-384) */
                    const int light_index = (int)__float_as_uint(lq.w);
                    const DLight& Lt = ENV ? S.env0 : S.lights[light_index];
                    const float4 md = W.pd_md ? q3 : W.ray[2 * (size_t)(p + W.n_paths) + 1];
                    Rgb inc(0.0f);
                    if (!(W.pd_occ ? (occ_word & 0xff00u) != 0u : W.occluded[p + W.n_paths] != 0)) inc = light_Le_env(Lt, V3(md.x, md.y, md.z));
                    if (!inc.is_black()) radiance = radiance + Rgb(q1.x, q1.y, q1.z) * inc * q0.w / q1.w;
                } else if (ps & PS_MIS) {
                    const int light_index = (int)__float_as_uint(lq.w);
                    const DLight& Lt = ENV ? S.env0 : S.lights[light_index];
                    const DHit mh = load_hit(W, p + W.n_paths);
                    const float4 mo = W.ray[2 * (size_t)(p + W.n_paths)], md = W.ray[2 * (size_t)(p + W.n_paths) + 1];
                    Rgb inc(0.0f);
                    if (mh.prim >= 0) {
                      if (!ENV) {                                /* ENV: no primitive carries an area light */
                        int h_mat, h_light; prim_mat_light(S, mh.prim, &h_mat, &h_light);
                        if (h_light >= 0 && h_light == light_index) {
                            DRay r0; r0.o = V3(mo.x, mo.y, mo.z); r0.d = V3(md.x, md.y, md.z); r0.t_max = FTN_INF; r0.time = 0.0f;
                            DSI s2; make_interaction(S, mh, r0, &s2);
                            inc = area_Le(Lt, s2.hit.n, -r0.d);
                        }
                      }
                    } else if (ENV || Lt.kind == LK_INFINITE) inc = light_Le_env(Lt, V3(md.x, md.y, md.z));
                    if (!inc.is_black()) radiance = radiance + Rgb(q1.x, q1.y, q1.z) * inc * q0.w / q1.w;
                }
                Rgb direct = Rgb(q2.x, q2.y, q2.z) * ((float)S.n_lights * radiance);
                L = L + direct;
                ps &= ~(PS_DIRECT | PS_SHADOW | PS_MIS | PS_MIS_ANY);
            }
            if (!(ps & PS_ALIVE)) {
                W.rad[p] = make_float4(L.r, L.g, L.b, 0.0f);     /* path finished: final radiance (the rest of its state is never read again) */
            } else {
                DRay ray0; ray0.o = V3(0.0f, 0.0f, 0.0f); ray0.d = V3(0.0f, 0.0f, 1.0f); ray0.t_max = FTN_INF; ray0.time = 0.0f;
                DHit h; h.t = FTN_INF; h.b0 = 0.0f; h.b1 = 0.0f; h.b2 = 0.0f; h.prim = -1;
                /* the light kernel of the classes without a BSDF mostly sees paths whose ray escaped: all such a path needs is that fact (hit_prim),
                 * unless it looks at the environment along the ray -- camera rays and specular bounces (path.rs:45-58).  Its ray and hit records are
                 * then not read at all */
                bool need_ray = true;
                if (MT == -2 && !first) { h.prim = W.hit_prim[p]; need_ray = h.prim >= 0 || (ps & PS_BOUNCE_MASK) == 0u || (ps & PS_SPECULAR) != 0u; }
                if (need_ray) {
                    const float4 ro = W.ray[2 * (size_t)(p)], rdv = W.ray[2 * (size_t)(p) + 1];
                    ray0.o = V3(ro.x, ro.y, ro.z); ray0.d = V3(rdv.x, rdv.y, rdv.z); ray0.t_max = rdv.w;
                    h = load_hit(W, p);
                }
                const bool hit = h.prim >= 0;
                uint32_t bounces = ps & PS_BOUNCE_MASK;
                const bool specular_bounce = (ps & PS_SPECULAR) != 0;
                DSI si;
                if (hit) make_interaction(S, h, ray0, &si);
                if (bounces == 0 || specular_bounce) {
                    if (hit) {
                        Rgb e(0.0f);
                        if (!ENV && si.light >= 0) e = area_Le(S.lights[si.light], si.hit.n, -ray0.d);
                        L = L + beta * e;
                    } else L = L + beta * (ENV ? Rgb(0.0f) + light_Le_env(S.env0, ray0.d) : scene_env_Le(S, ray0.d));
                }
                bool alive = true;
                uint32_t light_word = 0;
                if (!hit || bounces >= P.max_depth) alive = false;
                else {
                    Rng rng;
                    if (first) {      /* the stream behind the camera sample's five draws (p_film 2, p_lens 2, time 1: sampler/mod.rs:43-51), as k_wf_generate left it */
                        const uint32_t slot = p / W.samples, sidx = p % W.samples;
                        const DTile tile = P.tiles[slot >> 8];
                        rng.seed(indexed_key(P.seed, tile.x0 + (int)(slot & 15u), tile.y0 + (int)((slot >> 4) & 15u), W.first_sample + sidx));
                        (void)rng.next2(); (void)rng.next2(); (void)rng.next();
                    } else if (W.rng_replay) {   /* the same stream, re-created: seed of the sample's key, then the draws the path has made so far */
                        const uint32_t slot = p / W.samples, sidx = p % W.samples;
                        const DTile tile = P.tiles[slot >> 8];
                        rng.seed(indexed_key(P.seed, tile.x0 + (int)(slot & 15u), tile.y0 + (int)((slot >> 4) & 15u), W.first_sample + sidx));
                        rng.skip(ps >> PS_DRAWS_SHIFT);
                    } else { ulonglong2 a = W.rng01[p], b = W.rng23[p]; rng.s0 = a.x; rng.s1 = a.y; rng.s2 = b.x; rng.s3 = b.y; }
                    uint32_t n_draws = first ? 5u : (ps >> PS_DRAWS_SHIFT);
                    const int mat = si.mat;
                    if (MT == -2 || mat < 0) {
                        DRay nr = spawn_ray(si.hit, ray0.d);                       /* null bsdf: path.rs:77-81 */
                        W.ray[2 * (size_t)(p)] = make_float4(nr.o.x, nr.o.y, nr.o.z, 0.0f); W.ray[2 * (size_t)(p) + 1] = make_float4(nr.d.x, nr.d.y, nr.d.z, nr.t_max);
                        push_closest = true;
                    } else {
                        DBsdf B;
                        ftn_material mloc; const ftn_material* mp = &S.materials[mat];
                        if (TEX && material_is_textured(S, mat)) {
                            /* Texture::evaluate(si).  The differentials are the CAMERA ray's, handed on unchanged by the path integrator
                             * (path.rs:73): rebuilt here from the path's sample key instead of being carried in the path state. */
                            DRayDiff rd;
                            if (W.serial) {                      /* (no sample key in the tile-serial stream: k_wf_serial_advance stored them) */
                                const float4 d0 = W.dfd[p], d1 = W.dfd[(size_t)W.n_paths + p], d2 = W.dfd[2 * (size_t)W.n_paths + p];
                                rd.has = true; rd.rxo = V3(d0.x, d0.y, d0.z); rd.ryo = V3(d1.x, d1.y, d1.z); rd.ryd = V3(d2.x, d2.y, d2.z); rd.rxd = V3(d0.w, d1.w, d2.w);
                            } else {
                                const uint32_t slot = p / W.samples, sidx = p % W.samples;
                                const DTile tile = P.tiles[slot >> 8];
                                const int px = tile.x0 + (int)(slot & 15u), py = tile.y0 + (int)((slot >> 4) & 15u);
                                Rng crng; crng.seed(indexed_key(P.seed, px, py, W.first_sample + sidx));
                                const V2 j = crng.next2(); const V2 p_film((float)px + j.x, (float)py + j.y); const V2 p_lens = crng.next2(); const float time_u = crng.next();
                                const DRay cam = camera_ray(P.C, p_film, p_lens, time_u);
                                rd = camera_ray_diff(P.C, p_film, p_lens, cam, 1.0f / sqrtf((float)P.spp));
                            }
                            DSIX ex; DSI s2; make_interaction(S, h, ray0, &s2, &ex);
                            const DTexDiffs td = compute_tex_diffs(si.hit.p, si.hit.n, ex.dpdu, ex.dpdv, rd);
                            mloc = material_resolve(S, mat, ex.uv, td); mp = &mloc;
                        }
                        if (!make_bsdf<(MT >= 0 ? MT : -1)>(*mp, si, true, &B)) { err = FTN_ERR_UNSUPPORTED; alive = false; }
                        else {
                            if (bsdf_num(B, T_ALL & ~T_SPECULAR) > 0 && S.n_lights > 0) {
                                /* uniform_sample_one_light + first half of estimate_direct (integrator/mod.rs:289-329) */
                                const uint32_t nl = S.n_lights;
                                const uint32_t ln = (uint32_t)f2usize(fmin_(rng.next() * (float)nl, (float)(nl - 1)));
                                const V2 ul = rng.next2(), us = rng.next2();
                                n_draws += 5u;
                                const DLight& Lt = ENV ? S.env0 : S.lights[ln];
                                const uint32_t flags = T_ALL & ~T_SPECULAR;
                                const bool delta = !ENV && (Lt.kind == LK_POINT || Lt.kind == LK_DISTANT);
                                ps |= PS_DIRECT; light_word = ln;
                                Rgb ld(0.0f); float mis_w = 0.0f, mis_pdf = 1.0f; Rgb mis_f(0.0f); V3 mis_d(0.0f, 0.0f, 0.0f);
                                DLiSample ls = ENV ? light_sample_env(Lt, si.hit, ul) : light_sample(S, Lt, si.hit, ul);
                                if (ls.pdf > 0.0f && !ls.radiance.is_black()) {
                                    Rgb f = bsdf_f(B, si.wo, ls.wi, flags) * abs_dot(ls.wi, si.shading_n);
                                    float sp = bsdf_pdf(B, si.wo, ls.wi, flags);
                                    if (!f.is_black()) {
                                        DRay sr = spawn_ray_to_hit(si.hit, ls.p1);
                                        W.sh[2 * (size_t)(p)] = make_float4(sr.o.x, sr.o.y, sr.o.z, 0.0f); W.sh[2 * (size_t)(p) + 1] = make_float4(sr.d.x, sr.d.y, sr.d.z, sr.t_max);
                                        ld = delta ? (f * ls.radiance / ls.pdf) : (f * ls.radiance * power_heuristic(ls.pdf, sp) / ls.pdf);
                                        ps |= PS_SHADOW; push_shadow = true;
                                    }
                                }
                                if (!delta) {
                                    DScatter sc;
                                    if (bsdf_sample(B, si.wo, us, flags, &sc)) {
                                        Rgb f = sc.f * abs_dot(sc.wi, si.shading_n);
                                        if (!f.is_black()) {
                                            bool go = true;
                                            if (sc.type & T_SPECULAR) mis_w = 1.0f;
                                            else {
                                                float lp = ENV ? light_pdf_env(Lt, sc.wi) : light_pdf(S, Lt, si.hit, sc.wi);
                                                if (lp == 0.0f) go = false; else mis_w = power_heuristic(sc.pdf, lp);
                                            }
                                            if (go) {
                                                DRay mr = spawn_ray(si.hit, sc.wi);
                                                W.ray[2 * (size_t)(p + W.n_paths)] = make_float4(mr.o.x, mr.o.y, mr.o.z, 0.0f);
                                                W.ray[2 * (size_t)(p + W.n_paths) + 1] = make_float4(mr.d.x, mr.d.y, mr.d.z, mr.t_max);
                                                mis_f = f; mis_pdf = sc.pdf; mis_d = mr.d;
                                                if (W.mis_any && (ENV || Lt.kind == LK_INFINITE)) { ps |= PS_MIS_ANY; push_mis_any = true; } else { ps |= PS_MIS; push_mis = true; }
                                            }
                                        }
                                    }
                                }
                                if (W.pd) {
                                    float4* Q = W.pd + 4 * (size_t)p;
                                    Q[0] = make_float4(ld.r, ld.g, ld.b, mis_w); Q[1] = make_float4(mis_f.r, mis_f.g, mis_f.b, mis_pdf); Q[2] = make_float4(beta.r, beta.g, beta.b, 0.0f);
                                    if (W.pd_md) Q[3] = make_float4(mis_d.x, mis_d.y, mis_d.z, 0.0f);      /* w: the any-hit kernels' result bytes */
                                } else {
                                    W.pend0[p] = make_float4(ld.r, ld.g, ld.b, mis_w);
                                    W.pend1[p] = make_float4(mis_f.r, mis_f.g, mis_f.b, mis_pdf);
                                    W.pend2[p] = make_float4(beta.r, beta.g, beta.b, 0.0f);
                                }
                            }
                            /* sample the BSDF for the next direction (path.rs:67-76) */
                            DScatter bs;
                            const V2 u = rng.next2();
                            n_draws += 2u;
                            const bool ok = bsdf_sample(B, -ray0.d, u, T_ALL, &bs);
                            if (ok && !bs.f.is_black()) {
                                beta = beta * (bs.f * abs_dot(bs.wi, si.shading_n) / bs.pdf);
                                ps = (bs.type & T_SPECULAR) ? (ps | PS_SPECULAR) : (ps & ~PS_SPECULAR);
                                DRay nr = spawn_ray(si.hit, bs.wi);
                                /* Russian roulette (path.rs:84-91) */
                                if (beta.max_component() < P.rr_threshold && bounces > 3) {
                                    float q = fmax_(0.05f, 1.0f - beta.max_component());
                                    n_draws += 1u;
                                    if (rng.next() < q) alive = false; else beta = beta / (1.0f - q);
                                }
                                if (alive) {
                                    W.ray[2 * (size_t)(p)] = make_float4(nr.o.x, nr.o.y, nr.o.z, 0.0f); W.ray[2 * (size_t)(p) + 1] = make_float4(nr.d.x, nr.d.y, nr.d.z, nr.t_max);
                                    bounces += 1; push_closest = true;
                                }
                            } else alive = false;
                        }
                    }
                    if (W.rng_replay) ps = (ps & ((1u << PS_DRAWS_SHIFT) - 1u)) | (n_draws << PS_DRAWS_SHIFT);
                    else if (alive || W.serial) { W.rng01[p] = make_ulonglong2(rng.s0, rng.s1); W.rng23[p] = make_ulonglong2(rng.s2, rng.s3); }      /* (an ended path draws nothing more; the tile-serial stream goes on behind it) */
                }
                ps = (ps & ~(PS_BOUNCE_MASK | PS_ALIVE)) | (bounces & PS_BOUNCE_MASK) | (alive ? PS_ALIVE : 0u);
                push_active = alive || (ps & PS_DIRECT);        /* finished paths with a pending direct term come back once */
                if (W.br) {
                    if (push_active) { W.br[2 * (size_t)p] = make_float4(beta.r, beta.g, beta.b, __uint_as_float(ps)); W.br[2 * (size_t)p + 1] = make_float4(L.r, L.g, L.b, __uint_as_float(light_word)); }
                    else W.rad[p] = make_float4(L.r, L.g, L.b, 0.0f);
                } else {
                    if (push_active) W.beta[p] = make_float4(beta.r, beta.g, beta.b, __uint_as_float(ps));        /* (a retired path's throughput is never read again) */
                    W.rad[p] = make_float4(L.r, L.g, L.b, __uint_as_float(light_word));
                }
                active_entry = p | (alive ? (bounces >= P.max_depth ? WF_Q_DEPTH : 0u) : WF_Q_FIN);
            }
            if (W.serial && !push_active) W.ser_retired[p] = 1;      /* nothing pending: k_wf_serial_advance takes it from here */
        }
        {
            const bool pred[6] = {push_active, push_closest, push_mis, push_shadow, push_mis_any, push_mis_any};
            const uint32_t val[6] = {active_entry, p, p | WF_MIS_BIT, p, p | WF_MIS_BIT, 0u};
            uint32_t* const qs[6] = {out_q, W.q_closest, W.q_closest, W.q_shadow, W.q_shadow, nullptr};        /* the last "queue" only counts (CTR(10)): MIS rays traced as any-hit */
            uint32_t* const cs[6] = {out_count, &W.counters[CTR(2)], &W.counters[CTR(2)], &W.counters[CTR(3)], &W.counters[CTR(3)], &W.counters[CTR(10)]};
            block_push<6>(pred, val, qs, cs);
        }
    }
    if (err) atomicCAS(&P.stats->error, 0, err);
}

/* resident workgroups of a shade kernel per CU (registers, LDS), asked once per variant */
template <bool TEX, int MT, bool ENV> static unsigned shade_wgs_per_cu() {
    static const unsigned nb = [] { int v = 0; return (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_wf_shade<TEX, MT, ENV>, 256, 0) == hipSuccess && v >= 1) ? (unsigned)(v > 8 ? 8 : v) : 3u; }();
    return nb;
}

/* ------------------------------------------------------------------ DirectLightingIntegrator / WhittedIntegrator through the queues
 * (direct_lighting.rs:50-110, whitted.rs:20-75, specular_reflect / specular_transmit integrator/mod.rs:39-178)
 *
 * Their radiance is a NESTED product: Li(depth d) = local_d + f_d * Li(depth d + 1) * |cos_d| / pdf_d, evaluated innermost first.
 * A wavefront walks the chain forwards, so every depth keeps its terms -- dlA[d] = {local_d.rgb, pdf_d}, dlB[d] = {f_d.rgb, |cos_d|},
 * level-major SoA -- and the path is folded from its deepest level when it ends (dl_unwind: the same expressions, in the same order,
 * as the megakernel's direct_li and as the recursion they restate).  Only chains are supported, as in the megakernel: a BSDF with
 * both specular lobes (specular glass) reports FTN_ERR_UNSUPPORTED like the reference's todo!(); so does a primitive without material
 * (unimplemented!() at direct_lighting.rs:103) and a chain longer than WF_DL_MAX levels.
 * Per level: `local` = emitted radiance (direct lighting only) + the direct term.  DirectLightingIntegrator: uniform_sample_one_light,
 * the same deferred shadow / MIS rays as the path integrator (resolved when the path comes back after the traces), but called for every
 * BSDF (no "has non-specular lobes" test) and not weighted by a throughput.  WhittedIntegrator: every light once, one 2D sample and one
 * shadow ray each (ray l of path p in slot p + l * n_paths), up to WF_WH_MAX_LIGHTS lights; no MIS ray, no emitted term.
 * TEX (scenes with textures): texture lookups need the ray differentials, which these integrators carry along the specular chain
 * (specular_reflect updates them, integrator/mod.rs:58-84): level 0 rebuilds the camera ray's from the path's sample key (as k_wf_shade
 * does at every bounce), deeper levels read what the level above stored in WfBuffers::dfd.
 * The 2D sample of specular_transmit is drawn after the reflect recursion returns and never used by a supported material: with one
 * stream per camera sample (the indexed sampler, the only one the wavefront runs) skipping it changes nothing. */
#define WF_DL_MAX 8u
#define WF_WH_MAX_LIGHTS 32u     /* one bit per light in the path's pending-light word */
__device__ inline void dl_unwind(const RenderParams& P, const WfBuffers& W, uint32_t p, uint32_t depth, bool have_tail, Rgb tail) {
    Rgb li = have_tail ? tail : Rgb(0.0f);
    bool child = have_tail;
    for (int d = (int)depth - 1; d >= 0; d--) {
        const float4 a = W.dlA[(size_t)d * W.n_paths + p];
        Rgb r(a.x, a.y, a.z);
        if (child) { const float4 b = W.dlB[(size_t)d * W.n_paths + p]; r = r + Rgb(b.x, b.y, b.z) * li * b.w / a.w; }
        else if ((uint32_t)d + 1 < P.max_depth) r = r + Rgb(0.0f);                   /* radiance += specular_reflect (= 0) */
        if ((uint32_t)d + 1 < P.max_depth) r = r + Rgb(0.0f);                        /* radiance += specular_transmit (= 0) */
        li = r; child = true;
    }
    W.rad[p] = make_float4(li.r, li.g, li.b, 0.0f);
}

template <bool WHITTED, bool TEX>
__global__ void __launch_bounds__(256) k_wf_shade_dl(RenderParams P, WfBuffers W, int in_q) {
    const DScene& S = P.S;
    uint32_t cbase[WF_NCLASS + 1], ccnt[WF_NCLASS];
    cbase[0] = 0;
#pragma unroll
    for (int k = 0; k < WF_NCLASS; k++) { ccnt[k] = W.cls[CTR(k)]; cbase[k + 1] = cbase[k] + ((ccnt[k] + 255u) & ~255u); }
    const uint32_t count = cbase[WF_NCLASS];
    uint32_t* out_q = W.q_active[in_q ^ 1];
    uint32_t* out_count = &W.counters[CTR(in_q == 0 ? 1 : 0)];
    int err = 0;
    const uint32_t stride = gridDim.x * 256u;
    const uint32_t rounds = (count + stride - 1) / stride;
    const uint32_t nl = S.n_lights;
    for (uint32_t round = 0; round < rounds; round++) {
        const uint32_t qi = round * stride + blockIdx.x * 256u + threadIdx.x;
        bool have = false; uint32_t sorted_idx = 0;
        if (qi < count) {
            uint32_t c = 0, cb = 0, cc = ccnt[0];
#pragma unroll
            for (int k = 1; k < WF_NCLASS; k++) if (qi >= cbase[k]) { c = (uint32_t)k; cb = cbase[k]; cc = ccnt[k]; }
            have = qi - cb < cc;
            sorted_idx = c * W.seg_cap + (qi - cb);
        }
        bool push_active = false, push_closest = false, push_mis = false, push_mis_any = false;
        uint32_t sh_mask = 0;                                    /* shadow rays this path emits: bit l = the ray in slot p + l * n_paths (Whitted: one per light) */
        uint32_t p = 0, active_entry = 0;
        if (have) {
            p = W.q_sorted[sorted_idx];
            const float4 bq = W.beta[p], lq = W.rad[p];
            uint32_t ps = __float_as_uint(bq.w);
            uint32_t depth = ps & PS_BOUNCE_MASK;                 /* levels stored so far = index of the level shaded now */
            /* ---- finish the direct term of level depth - 1 (its rays have been traced) */
            if (ps & PS_DIRECT) {
                const uint32_t lv = depth - 1u;
                float4 a = W.dlA[(size_t)lv * W.n_paths + p];
                Rgb rad(a.x, a.y, a.z);
                if (WHITTED) {                                   /* whitted.rs:42-58: radiance += f * Li * |cos| / pdf for every unoccluded light, in light order */
                    const uint32_t mask = __float_as_uint(lq.w);
                    for (uint32_t l = 0; l < nl; l++) if ((mask >> l) & 1u) {
                        const float4 t = W.whT[(size_t)l * W.n_paths + p];
                        if (!W.occluded[p + (size_t)l * W.n_paths]) rad = rad + Rgb(t.x, t.y, t.z);
                    }
                } else {                                         /* uniform_sample_one_light = n_lights * estimate_direct (integrator/mod.rs:289-395) */
                    const float4 q0 = W.pend0[p], q1 = W.pend1[p];
                    Rgb radiance(0.0f);
                    if ((ps & PS_SHADOW) && !W.occluded[p]) radiance = radiance + Rgb(q0.x, q0.y, q0.z);
                    const int light_index = (int)__float_as_uint(lq.w);
                    const DLight& Lt = S.lights[light_index];
                    if (ps & PS_MIS_ANY) {
                        const float4 md = W.ray[2 * (size_t)(p + W.n_paths) + 1];
                        Rgb inc(0.0f);
                        if (!W.occluded[p + W.n_paths]) inc = light_Le_env(Lt, V3(md.x, md.y, md.z));
                        if (!inc.is_black()) radiance = radiance + Rgb(q1.x, q1.y, q1.z) * inc * q0.w / q1.w;
                    } else if (ps & PS_MIS) {
                        const DHit mh = load_hit(W, p + W.n_paths);
                        const float4 mo = W.ray[2 * (size_t)(p + W.n_paths)], md = W.ray[2 * (size_t)(p + W.n_paths) + 1];
                        Rgb inc(0.0f);
                        if (mh.prim >= 0) {
                            int h_mat, h_light; prim_mat_light(S, mh.prim, &h_mat, &h_light);
                            if (h_light >= 0 && h_light == light_index) {
                                DRay r0; r0.o = V3(mo.x, mo.y, mo.z); r0.d = V3(md.x, md.y, md.z); r0.t_max = FTN_INF; r0.time = 0.0f;
                                DSI s2; make_interaction(S, mh, r0, &s2);
                                inc = area_Le(Lt, s2.hit.n, -r0.d);
                            }
                        } else if (Lt.kind == LK_INFINITE) inc = light_Le_env(Lt, V3(md.x, md.y, md.z));
                        if (!inc.is_black()) radiance = radiance + Rgb(q1.x, q1.y, q1.z) * inc * q0.w / q1.w;
                    }
                    rad = rad + (float)nl * radiance;
                }
                a.x = rad.r; a.y = rad.g; a.z = rad.b;
                W.dlA[(size_t)lv * W.n_paths + p] = a;
                ps &= ~(PS_DIRECT | PS_SHADOW | PS_MIS | PS_MIS_ANY);
            }
            if (!(ps & PS_ALIVE)) {
                dl_unwind(P, W, p, depth, false, Rgb(0.0f));       /* the chain ended at its last level: fold it */
                W.beta[p] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(ps));
            } else {
                const float4 ro = W.ray[2 * (size_t)(p)], rdv = W.ray[2 * (size_t)(p) + 1];
                DRay ray0; ray0.o = V3(ro.x, ro.y, ro.z); ray0.d = V3(rdv.x, rdv.y, rdv.z); ray0.t_max = rdv.w; ray0.time = 0.0f;
                const DHit h = load_hit(W, p);
                bool alive = false; uint32_t light_word = 0; bool ended = false;
                if (h.prim < 0) { dl_unwind(P, W, p, depth, true, scene_env_Le(S, ray0.d)); ended = true; }      /* None => environment_emitted_radiance */
                else if (depth >= WF_DL_MAX) { err = FTN_ERR_UNSUPPORTED; ended = true; }
                else {
                    DSI si; make_interaction(S, h, ray0, &si);
                    DBsdf B;
                    ftn_material mloc; const ftn_material* mp = si.mat >= 0 ? &S.materials[si.mat] : nullptr;
                    DSIX ex; DTexDiffs td; float bsdf_eta = 1.0f; DRayDiff rd; rd.has = false;
                    if (TEX && mp) {                              /* as direct_li<TEX> (ftn_kernels.hip): differentials of this level, textured parameters resolved per hit */
                        if (depth == 0u) {
                            const uint32_t slot = p / W.samples, sidx = p % W.samples;
                            const DTile tile = P.tiles[slot >> 8];
                            const int px = tile.x0 + (int)(slot & 15u), py = tile.y0 + (int)((slot >> 4) & 15u);
                            Rng crng; crng.seed(indexed_key(P.seed, px, py, W.first_sample + sidx));
                            const V2 j = crng.next2(); const V2 p_film((float)px + j.x, (float)py + j.y); const V2 p_lens = crng.next2(); const float time_u = crng.next();
                            const DRay cam = camera_ray(P.C, p_film, p_lens, time_u);
                            rd = camera_ray_diff(P.C, p_film, p_lens, cam, 1.0f / sqrtf((float)P.spp));
                        } else {
                            const float4 d0 = W.dfd[p], d1 = W.dfd[(size_t)W.n_paths + p], d2 = W.dfd[2 * (size_t)W.n_paths + p];
                            rd.has = true; rd.rxo = V3(d0.x, d0.y, d0.z); rd.ryo = V3(d1.x, d1.y, d1.z); rd.ryd = V3(d2.x, d2.y, d2.z); rd.rxd = V3(d0.w, d1.w, d2.w);
                        }
                        DSI s2; make_interaction(S, h, ray0, &s2, &ex);
                        td = compute_tex_diffs(si.hit.p, si.hit.n, ex.dpdu, ex.dpdv, rd);
                        if (material_is_textured(S, si.mat)) { mloc = material_resolve(S, si.mat, ex.uv, td); mp = &mloc; }
                        if (mp->type == FTN_MAT_GLASS) bsdf_eta = mp->s0;
                    }
                    if (!mp || !make_bsdf<-1>(*mp, si, false, &B)) { err = FTN_ERR_UNSUPPORTED; ended = true; }
                    else {
                        Rng rng; { ulonglong2 a = W.rng01[p], b = W.rng23[p]; rng.s0 = a.x; rng.s1 = a.y; rng.s2 = b.x; rng.s3 = b.y; }
                        Rgb rad(0.0f);
                        if (WHITTED) {
                            uint32_t mask = 0;
                            if (nl > WF_WH_MAX_LIGHTS) err = FTN_ERR_UNSUPPORTED;
                            else for (uint32_t l = 0; l < nl; l++) {
                                const DLight& Lt = S.lights[l];
                                DLiSample ls = light_sample(S, Lt, si.hit, rng.next2());
                                if (ls.radiance.is_black() || ls.pdf == 0.0f) continue;
                                Rgb f = bsdf_f(B, si.wo, ls.wi, T_ALL);
                                if (!f.is_black()) {
                                    DRay sr = spawn_ray_to_hit(si.hit, ls.p1);
                                    const size_t slot = p + (size_t)l * W.n_paths;
                                    W.sh[2 * slot] = make_float4(sr.o.x, sr.o.y, sr.o.z, 0.0f); W.sh[2 * slot + 1] = make_float4(sr.d.x, sr.d.y, sr.d.z, sr.t_max);
                                    const Rgb term = f * ls.radiance * abs_dot(ls.wi, si.shading_n) / ls.pdf;
                                    W.whT[slot] = make_float4(term.r, term.g, term.b, 0.0f);
                                    mask |= 1u << l;
                                }
                            }
                            if (mask) { ps |= PS_DIRECT; light_word = mask; sh_mask = mask; }
                        } else {
                            if (si.light >= 0) rad = rad + area_Le(S.lights[si.light], si.hit.n, si.wo);      /* radiance += intersect.emitted_radiance(wo) */
                            else rad = rad + Rgb(0.0f);
                            if (nl > 0) {                                     /* uniform_sample_one_light + first half of estimate_direct */
                                const uint32_t ln = (uint32_t)f2usize(fmin_(rng.next() * (float)nl, (float)(nl - 1)));
                                const V2 ul = rng.next2(), us = rng.next2();
                                const DLight& Lt = S.lights[ln];
                                const uint32_t flags = T_ALL & ~T_SPECULAR;
                                const bool delta = Lt.kind == LK_POINT || Lt.kind == LK_DISTANT;
                                ps |= PS_DIRECT; light_word = ln;
                                Rgb ld(0.0f); float mis_w = 0.0f, mis_pdf = 1.0f; Rgb mis_f(0.0f);
                                DLiSample ls = light_sample(S, Lt, si.hit, ul);
                                if (ls.pdf > 0.0f && !ls.radiance.is_black()) {
                                    Rgb f = bsdf_f(B, si.wo, ls.wi, flags) * abs_dot(ls.wi, si.shading_n);
                                    float sp = bsdf_pdf(B, si.wo, ls.wi, flags);
                                    if (!f.is_black()) {
                                        DRay sr = spawn_ray_to_hit(si.hit, ls.p1);
                                        W.sh[2 * (size_t)(p)] = make_float4(sr.o.x, sr.o.y, sr.o.z, 0.0f); W.sh[2 * (size_t)(p) + 1] = make_float4(sr.d.x, sr.d.y, sr.d.z, sr.t_max);
                                        ld = delta ? (f * ls.radiance / ls.pdf) : (f * ls.radiance * power_heuristic(ls.pdf, sp) / ls.pdf);
                                        ps |= PS_SHADOW; sh_mask = 1u;
                                    }
                                }
                                if (!delta) {
                                    DScatter sc;
                                    if (bsdf_sample(B, si.wo, us, flags, &sc)) {
                                        Rgb f = sc.f * abs_dot(sc.wi, si.shading_n);
                                        if (!f.is_black()) {
                                            bool go = true;
                                            if (sc.type & T_SPECULAR) mis_w = 1.0f;
                                            else { float lp = light_pdf(S, Lt, si.hit, sc.wi); if (lp == 0.0f) go = false; else mis_w = power_heuristic(sc.pdf, lp); }
                                            if (go) {
                                                DRay mr = spawn_ray(si.hit, sc.wi);
                                                W.ray[2 * (size_t)(p + W.n_paths)] = make_float4(mr.o.x, mr.o.y, mr.o.z, 0.0f);
                                                W.ray[2 * (size_t)(p + W.n_paths) + 1] = make_float4(mr.d.x, mr.d.y, mr.d.z, mr.t_max);
                                                mis_f = f; mis_pdf = sc.pdf;
                                                if (W.mis_any && Lt.kind == LK_INFINITE) { ps |= PS_MIS_ANY; push_mis_any = true; } else { ps |= PS_MIS; push_mis = true; }
                                            }
                                        }
                                    }
                                }
                                W.pend0[p] = make_float4(ld.r, ld.g, ld.b, mis_w);
                                W.pend1[p] = make_float4(mis_f.r, mis_f.g, mis_f.b, mis_pdf);
                            }
                        }
                        float4 lvA = make_float4(rad.r, rad.g, rad.b, 1.0f), lvB = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        if (depth + 1u < P.max_depth) {
                            /* specular_reflect: the 2D sample is drawn before the match (mod.rs:52) */
                            const V2 ur = rng.next2();
                            DScatter sr;
                            const bool has_r = bsdf_sample(B, si.wo, ur, T_REFL | T_SPECULAR, &sr) && !(abs_dot(sr.wi, si.shading_n) == 0.0f);
                            const bool has_t = bsdf_num(B, T_TRANS | T_SPECULAR) > 0;
                            if (has_t) err = FTN_ERR_UNSUPPORTED;              /* specular transmission: glass.rs:66 todo!() */
                            else if (has_r) {
                                lvA.w = sr.pdf; lvB = make_float4(sr.f.r, sr.f.g, sr.f.b, fabsf(dot(sr.wi, si.shading_n)));
                                if (TEX) {                            /* the reflected ray's differentials, for the next level's lookups (mod.rs:58-84) */
                                    rd = specular_diff(true, rd, si.hit.p, si.wo, sr.wi, si.shading_n, ex.dndu, ex.dndv, td, bsdf_eta);
                                    W.dfd[p] = make_float4(rd.rxo.x, rd.rxo.y, rd.rxo.z, rd.rxd.x); W.dfd[(size_t)W.n_paths + p] = make_float4(rd.ryo.x, rd.ryo.y, rd.ryo.z, rd.rxd.y);
                                    W.dfd[2 * (size_t)W.n_paths + p] = make_float4(rd.ryd.x, rd.ryd.y, rd.ryd.z, rd.rxd.z);
                                }
                                DRay nr = spawn_ray(si.hit, sr.wi);
                                W.ray[2 * (size_t)(p)] = make_float4(nr.o.x, nr.o.y, nr.o.z, 0.0f); W.ray[2 * (size_t)(p) + 1] = make_float4(nr.d.x, nr.d.y, nr.d.z, nr.t_max);
                                alive = true; push_closest = true;
                            }
                        }
                        W.dlA[(size_t)depth * W.n_paths + p] = lvA; W.dlB[(size_t)depth * W.n_paths + p] = lvB;
                        depth += 1u;
                        W.rng01[p] = make_ulonglong2(rng.s0, rng.s1); W.rng23[p] = make_ulonglong2(rng.s2, rng.s3);
                        if (!alive && !(ps & PS_DIRECT)) { dl_unwind(P, W, p, depth, false, Rgb(0.0f)); ended = true; }      /* nothing pending: fold now */
                    }
                }
                if (err && !ended) ended = true;
                if (ended) { alive = false; ps &= ~(PS_DIRECT | PS_SHADOW | PS_MIS | PS_MIS_ANY); sh_mask = 0; push_closest = push_mis = push_mis_any = false; if (err) W.rad[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
                ps = (ps & ~(PS_BOUNCE_MASK | PS_ALIVE)) | (depth & PS_BOUNCE_MASK) | (alive ? PS_ALIVE : 0u);
                W.beta[p] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(ps));
                if (!ended) W.rad[p] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(light_word));
                push_active = alive || (ps & PS_DIRECT);
                active_entry = p | (alive ? 0u : WF_Q_FIN);
            }
        }
        {
            const bool pred[5] = {push_active, push_closest, push_mis, push_mis_any, push_mis_any};
            const uint32_t val[5] = {active_entry, p, p | WF_MIS_BIT, p | WF_MIS_BIT, 0u};
            uint32_t* const qs[5] = {out_q, W.q_closest, W.q_closest, W.q_shadow, nullptr};
            uint32_t* const cs[5] = {out_count, &W.counters[CTR(2)], &W.counters[CTR(2)], &W.counters[CTR(3)], &W.counters[CTR(10)]};
            block_push<5>(pred, val, qs, cs);
        }
        /* shadow rays: slot p + l * n_paths (direct lighting: l = 0 only) -- queue entries are slots; the any-hit kernels read W.sh[2 * slot] */
        for (uint32_t l = 0; l < (WHITTED ? nl : 1u); l++) {
            const bool on = ((sh_mask >> l) & 1u) != 0;
            const bool pred[1] = {on};
            const uint32_t val[1] = {p + l * W.n_paths};
            uint32_t* const qs[1] = {W.q_shadow};
            uint32_t* const cs[1] = {&W.counters[CTR(3)]};
            if (__syncthreads_or(on ? 1 : 0)) block_push<1>(pred, val, qs, cs);
        }
    }
    if (err) atomicCAS(&P.stats->error, 0, err);
}

/* counters housekeeping between kernels (one tiny launch instead of host round trips) */
__global__ void k_wf_reset(WfBuffers W, int mode, int in_q, DevStats* stats) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (mode == 0) {                                                                         /* new pass: generate fills both queues densely */
        for (int i = 0; i < 64; i++) W.counters[CTR(i)] = 0;
#ifdef FTN_DRAIN_PROBE
        for (int k = 0; k < 32; k++) W.counters[CTR(13) + k] = 0;
#endif
        W.counters[CTR(0)] = W.samples * W.valid_per_sample; W.counters[CTR(2)] = W.samples * W.valid_per_sample;
        stats->camera_samples += (unsigned long long)W.samples * W.valid_per_sample;
    } else if (mode == 1) {                                                                  /* before shade */
        W.counters[CTR(2)] = 0; W.counters[CTR(3)] = 0; W.counters[CTR(in_q == 0 ? 1 : 0)] = 0; W.counters[CTR(11)] = 0;
        for (int i = 16; i < 32; i++) W.counters[CTR(i)] = 0;                                /* the per-XCD queue heads of the two trace kernels */
        for (int i = 32; i < 56; i++) W.counters[CTR(i)] = 0;                                /* exception queues of the four-box kernels: counts and heads */
    } else if (mode == 2) {                                                                  /* before classify */
        for (int i = 0; i < WF_NCLASS; i++) W.cls[CTR(i)] = 0;
    } else if (mode == 3) {                                                                  /* tile-serial pass: the queues start empty, k_wf_serial_advance fills them */
        for (int i = 0; i < 64; i++) W.counters[CTR(i)] = 0;
    }
}

/* ------------------------------------------------------------------ accumulate: add_sample_to_tile in sample order */
struct FilmCtxW { int crop[4]; int tpb[4]; int sb[4]; float radius[2]; };
#define WF_OWN_SERIAL (-2147483647 - 1)
__device__ inline void wf_film_add(const RenderParams& P, const FilmCtxW& F, V2 p_film, Rgb L, int own_x, int own_y, float4* acc, uint32_t* spill, uint32_t* bc_writes) {
    float pdx = p_film.x - 0.5f, pdy = p_film.y - 0.5f;
    int p0x = f2i_sat(ceilf(pdx - F.radius[0])), p0y = f2i_sat(ceilf(pdy - F.radius[1]));
    int p1x = f2i_sat(floorf(pdx + F.radius[0])) + 1, p1y = f2i_sat(floorf(pdy + F.radius[1])) + 1;
    p0x = max(p0x, F.tpb[0]); p0y = max(p0y, F.tpb[1]); p1x = min(p1x, F.tpb[2]); p1y = min(p1y, F.tpb[3]);
    const Rgb contrib = L * 1.0f * 1.0f;                     /* radiance * sample_weight * filter_weight (box: 1.0) */
    int touched = 0;
    const size_t width = (size_t)(F.crop[2] - F.crop[0]);
    for (int y = p0y; y < p1y; y++)
        for (int x = p0x; x < p1x; x++) {
            touched++;
            if (x == own_x && y == own_y) { acc->x += contrib.r; acc->y += contrib.g; acc->z += contrib.b; acc->w += 1.0f; continue; }
            const bool in_tile = x >= F.sb[0] && x < F.sb[2] && y >= F.sb[1] && y < F.sb[3];
            if (in_tile && own_x == WF_OWN_SERIAL) {         /* tile-serial: the tile's one writer adds in-tile samples straight into A, in stream order (film_add, ftn_kernels.hip) */
                float4* a = P.accA + ((size_t)(y - F.crop[1]) * width + (size_t)(x - F.crop[0]));
                float4 v = *a; v.x += contrib.r; v.y += contrib.g; v.z += contrib.b; v.w += 1.0f; *a = v;
                continue;
            }
            float* f = reinterpret_cast<float*>((in_tile ? P.accB : P.accC) + ((size_t)(y - F.crop[1]) * width + (size_t)(x - F.crop[0])));
            atomicAdd(f + 0, contrib.r); atomicAdd(f + 1, contrib.g); atomicAdd(f + 2, contrib.b); atomicAdd(f + 3, 1.0f);
            (*bc_writes)++;
        }
    if (touched != 1) (*spill)++;
}
/* One thread per pixel slot adds its samples in sample order (film.rs:127-130 is order dependent in the last bit).  The samples of a
 * slot are neighbours in memory (path id = slot * samples + s), so a thread reading its own run would touch one cache line per lane and
 * load; the workgroup stages 8 samples of its 256 slots through LDS with coalesced loads instead. */
#define WF_ACC_CHUNK 8u
__global__ void __launch_bounds__(256) k_wf_accumulate(RenderParams P, WfBuffers W) {
    __shared__ float4 s_rad[256 * WF_ACC_CHUNK];
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    uint32_t spill = 0, bc = 0, cam = 0; int err = 0;
    bool valid = false, in_crop = false; int px = 0, py = 0; size_t ai = 0;
    FilmCtxW F; float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (slot < W.n_slots) {
        const DTile tile = P.tiles[slot >> 8];
        px = tile.x0 + (int)(slot & 15u); py = tile.y0 + (int)((slot >> 4) & 15u);
        if (px < tile.x1 && py < tile.y1) {
            valid = true;
            for (int i = 0; i < 4; i++) F.crop[i] = P.crop[i];
            F.sb[0] = tile.x0; F.sb[1] = tile.y0; F.sb[2] = tile.x1; F.sb[3] = tile.y1; F.radius[0] = P.radius[0]; F.radius[1] = P.radius[1];
            int p0x = f2i_sat(ceilf((float)tile.x0 - 0.5f - P.radius[0])), p0y = f2i_sat(ceilf((float)tile.y0 - 0.5f - P.radius[1]));
            int p1x = f2i_sat(ceilf((float)tile.x1 - 0.5f + P.radius[0] + 1.0f)), p1y = f2i_sat(ceilf((float)tile.y1 - 0.5f - P.radius[1] + 1.0f));
            F.tpb[0] = max(p0x, P.crop[0]); F.tpb[1] = max(p0y, P.crop[1]); F.tpb[2] = min(p1x, P.crop[2]); F.tpb[3] = min(p1y, P.crop[3]);
            in_crop = px >= P.crop[0] && px < P.crop[2] && py >= P.crop[1] && py < P.crop[3];
            ai = in_crop ? ((size_t)(py - P.crop[1]) * (size_t)(P.crop[2] - P.crop[0]) + (size_t)(px - P.crop[0])) : 0;
            if (in_crop) acc = P.accA[ai];
        }
    }
    const size_t block_first = (size_t)blockIdx.x * 256u * W.samples;          /* first path of this workgroup's 256 slots */
    for (uint32_t s0 = 0; s0 < W.samples; s0 += WF_ACC_CHUNK) {
        const uint32_t n = W.samples - s0 < WF_ACC_CHUNK ? W.samples - s0 : WF_ACC_CHUNK;
        for (uint32_t e = threadIdx.x; e < 256u * n; e += 256u) {                /* n consecutive samples of slot e / n */
            const uint32_t sl = e / n, k = e - sl * n;
            const size_t p = block_first + (size_t)sl * W.samples + s0 + k;
            if (p < W.n_paths) s_rad[sl * WF_ACC_CHUNK + k] = W.rad[p];
        }
        __syncthreads();
        if (valid) {
            for (uint32_t k = 0; k < n; k++) {
                const float4 l = s_rad[threadIdx.x * WF_ACC_CHUNK + k];
                Rgb L(l.x, l.y, l.z);
                if (L.has_nans()) err = FTN_ERR_NAN_RADIANCE;
                /* the sample's film position: the first two draws of its stream, exactly as k_wf_generate made them (sampler/mod.rs:41-48) */
                Rng crng; crng.seed(indexed_key(P.seed, px, py, W.first_sample + s0 + k));
                const V2 j = crng.next2();
                wf_film_add(P, F, V2((float)px + j.x, (float)py + j.y), L, in_crop ? px : (-2147483647), py, &acc, &spill, &bc);
                cam++;
            }
        }
        __syncthreads();
    }
    if (valid && in_crop) P.accA[ai] = acc;
    (void)cam;                                               /* camera_samples is known in closed form: added once by k_wf_reset */
    if (spill) atomicAdd(&P.stats->spill_samples, (unsigned long long)spill);       /* rare */
    if (bc) atomicAdd(&P.stats->bc_writes, (unsigned long long)bc);
    if (err) atomicCAS(&P.stats->error, 0, err);
}

/* ------------------------------------------------------------------ the reference's RandomSampler on the queues (render_tile, integrator/mod.rs:229-281, with
 * sampler.clone_with_seed(tile_id), random.rs:61-67): one thread per tile.  init: seeds the tile's stream and starts its first camera sample.
 * Afterwards, once per bounce round: a tile whose path was marked retired adds that radiance to the film -- samples of a tile in stream
 * order, in-tile pixels by this single writer as k_render_serial does -- and draws the next camera sample (pixel by pixel, row-major,
 * spp samples each) from the stream position the finished path left. */
template <bool TEX>
__global__ void __launch_bounds__(256) k_wf_serial_advance(RenderParams P, WfBuffers W, int out_q, int init) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    bool push = false;
    uint32_t spill = 0, bc = 0, cam = 0; int err = 0;
    if (t < W.n_paths) {
        const DTile tile = P.tiles[t];
        const uint32_t tw = (uint32_t)(tile.x1 - tile.x0), npix = tw * (uint32_t)(tile.y1 - tile.y0);
        Rng rng; uint2 cur = make_uint2(0u, 0u); bool start = false;
        if (init) { rng.seed(tile.tile_id); start = npix > 0u && P.spp > 0u; W.ser_retired[t] = 0; }
        else if (W.ser_retired[t]) {
            W.ser_retired[t] = 0;
            { const ulonglong2 a = W.rng01[t], b = W.rng23[t]; rng.s0 = a.x; rng.s1 = a.y; rng.s2 = b.x; rng.s3 = b.y; }
            cur = W.ser_cursor[t];
            const float4 l = W.rad[t]; const float2 pf = W.ser_pfilm[t];
            Rgb L(l.x, l.y, l.z);
            if (L.has_nans()) err = FTN_ERR_NAN_RADIANCE;
            FilmCtxW F;
            for (int i = 0; i < 4; i++) F.crop[i] = P.crop[i];
            F.sb[0] = tile.x0; F.sb[1] = tile.y0; F.sb[2] = tile.x1; F.sb[3] = tile.y1; F.radius[0] = P.radius[0]; F.radius[1] = P.radius[1];
            const int p0x = f2i_sat(ceilf((float)tile.x0 - 0.5f - P.radius[0])), p0y = f2i_sat(ceilf((float)tile.y0 - 0.5f - P.radius[1]));
            const int p1x = f2i_sat(ceilf((float)tile.x1 - 0.5f + P.radius[0] + 1.0f)), p1y = f2i_sat(ceilf((float)tile.y1 - 0.5f - P.radius[1] + 1.0f));
            F.tpb[0] = max(p0x, P.crop[0]); F.tpb[1] = max(p0y, P.crop[1]); F.tpb[2] = min(p1x, P.crop[2]); F.tpb[3] = min(p1y, P.crop[3]);
            float4 none = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            wf_film_add(P, F, V2(pf.x, pf.y), L, WF_OWN_SERIAL, 0, &none, &spill, &bc);
            cam = 1;
            if (++cur.y == P.spp) { cur.y = 0; cur.x++; }
            start = cur.x < npix;
        }
        if (start) {
            const int px = tile.x0 + (int)(cur.x % tw), py = tile.y0 + (int)(cur.x / tw);
            const V2 j = rng.next2();
            const V2 p_film((float)px + j.x, (float)py + j.y);
            const V2 p_lens = rng.next2();
            const float time_u = rng.next();
            const DRay ray = camera_ray(P.C, p_film, p_lens, time_u);
            W.ray[2 * (size_t)t] = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.0f);
            W.ray[2 * (size_t)t + 1] = make_float4(ray.d.x, ray.d.y, ray.d.z, ray.t_max);
            W.beta[t] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(PS_ALIVE));
            W.rad[t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            W.rng01[t] = make_ulonglong2(rng.s0, rng.s1); W.rng23[t] = make_ulonglong2(rng.s2, rng.s3);
            W.ser_cursor[t] = cur; W.ser_pfilm[t] = make_float2(p_film.x, p_film.y);
            if (TEX) {                                           /* generate_ray_differential + scale_differentials (mod.rs:252-254) */
                const DRayDiff rd = camera_ray_diff(P.C, p_film, p_lens, ray, 1.0f / sqrtf((float)P.spp));
                W.dfd[t] = make_float4(rd.rxo.x, rd.rxo.y, rd.rxo.z, rd.rxd.x); W.dfd[(size_t)W.n_paths + t] = make_float4(rd.ryo.x, rd.ryo.y, rd.ryo.z, rd.rxd.y);
                W.dfd[2 * (size_t)W.n_paths + t] = make_float4(rd.ryd.x, rd.ryd.y, rd.ryd.z, rd.rxd.z);
            }
            push = true;
        }
    }
    {
        const bool pred[2] = {push, push};
        const uint32_t val[2] = {t | (P.max_depth == 0 ? WF_Q_DEPTH : 0u), t};
        uint32_t* const qs[2] = {W.q_active[out_q], W.q_closest};
        uint32_t* const cs[2] = {&W.counters[CTR(out_q == 0 ? 0 : 1)], &W.counters[CTR(2)]};
        block_push<2>(pred, val, qs, cs);
    }
    if (cam) atomicAdd(&P.stats->camera_samples, (unsigned long long)cam);
    if (spill) atomicAdd(&P.stats->spill_samples, (unsigned long long)spill);
    if (bc) atomicAdd(&P.stats->bc_writes, (unsigned long long)bc);
    if (err) atomicCAS(&P.stats->error, 0, err);
}

/* ================================================================== host driver */
static thread_local std::string g_wf_err;
const char* wavefront_error() { return g_wf_err.c_str(); }

/* ------------------------------------------------------------------ coherence sort of the secondary-ray queues
 * After a bounce the queues hold rays in shading order: origins of neighbouring entries are still close, directions are not.
 * Sorting a queue by (Morton code of the origin's cell in the world box, direction octant) puts rays that walk the same part of
 * the BVH into the same wave, so a node record fetched by one lane is an L1 / L2 hit for its neighbours.  Measured on the config-5
 * scene: secondary closest-hit launches 18.1 -> 14.1 ms per step, shadow launches 8.4 -> 7.6 ms, for ~1.3 ms of sorting
 * (key kernel + rocPRIM radix sort of (key, ray id) pairs, 24 significant bits).  The order in which rays are traced does not
 * change any result (every ray writes its own hit record), so parity is untouched.  FTN_WF_SORT=0 disables it. */
__device__ inline uint32_t spread3(uint32_t v) {   /* 10 bits -> every third bit */
    v &= 0x3ffu; v = (v | (v << 16)) & 0x030000ffu; v = (v | (v << 8)) & 0x0300f00fu; v = (v | (v << 4)) & 0x030c30c3u; v = (v | (v << 2)) & 0x09249249u; return v;
}
template <bool ANY>
__global__ void __launch_bounds__(256) k_wf_ray_keys(DScene S, WfBuffers W, const uint32_t* __restrict__ queue, uint32_t count, uint32_t* __restrict__ keys, uint32_t bits) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    float4 a, b; uint32_t slot;
    load_queued_ray<ANY>(W, queue[i], &a, &b, &slot);
    const float ex = S.root_hi[0] - S.root_lo[0], ey = S.root_hi[1] - S.root_lo[1], ez = S.root_hi[2] - S.root_lo[2];
    const float sc = (float)(1u << bits);
    const float fx = fminf(fmaxf((a.x - S.root_lo[0]) / ex, 0.0f), 0.999f) * sc, fy = fminf(fmaxf((a.y - S.root_lo[1]) / ey, 0.0f), 0.999f) * sc, fz = fminf(fmaxf((a.z - S.root_lo[2]) / ez, 0.0f), 0.999f) * sc;
    const uint32_t m = spread3((uint32_t)fx) | (spread3((uint32_t)fy) << 1) | (spread3((uint32_t)fz) << 2);       /* NaN / inf -> cell 0: still a valid key */
    const uint32_t oct = (b.x < 0.0f ? 1u : 0u) | (b.y < 0.0f ? 2u : 0u) | (b.z < 0.0f ? 4u : 0u);
    keys[i] = (m << 3) | oct;
}

struct WavefrontState {
    void* sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
    size_t cap_paths = 0;
    void* mem[40]; int n_mem = 0;
    WfBuffers W;
    float4* br = nullptr; float4* pd = nullptr;          /* WfBuffers::br, WfBuffers::pd */
    hipEvent_t ev[64]; int n_ev = 0;
    hipStream_t side = nullptr; hipEvent_t ev_ready = nullptr, ev_side = nullptr;     /* the any-hit launches run beside the closest-hit ones */
    uint32_t* drain_sig = nullptr; uint32_t drain_seq = 0;                              /* signal memory for hipStreamWaitValue32 (NULL: not supported) */
    uint32_t* host_counters = nullptr;    /* pinned */
    int n_cu = 256;
    /* four-box traversal (ftn_trace4.hip): launch plan of the current call and the global spill areas behind the LDS stacks */
    /* buffers of the direct-lighting / Whitted mode (grow-only): level terms, and shadow-ray records / results / queue sized for one ray per light */
    void* dl_mem[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; size_t dl_paths = 0; uint32_t dl_levels = 0, dl_slots = 0; bool dl_tex = false;
    Trace4Plan t4; bool t4_on = false, t8_on = false;
    void* ser_mem[4] = {nullptr, nullptr, nullptr, nullptr}; size_t ser_paths = 0; bool ser_tex = false; uint32_t* ser_host = nullptr;      /* tile-serial sampler on the queues: cursor, film position, retired flag, differentials */
    void* t4_spill_c = nullptr; void* t4_spill_a = nullptr; size_t t4_spill_c_bytes = 0, t4_spill_a_bytes = 0;
};
#define WF_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_wf_err = std::string(#expr ": ") + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? FTN_ERR_OUT_OF_MEMORY : FTN_ERR_NO_DEVICE; } } while (0)

static void wf_free(WavefrontState* st) { for (int i = 0; i < st->n_mem; i++) (void)hipFree(st->mem[i]); st->n_mem = 0; st->cap_paths = 0; }
void wavefront_destroy(WavefrontState* st) {
    if (!st) return;
    wf_free(st);
    if (st->sort_tmp) (void)hipFree(st->sort_tmp);
    for (int i = 0; i < st->n_ev; i++) (void)hipEventDestroy(st->ev[i]);
    if (st->ev_ready) (void)hipEventDestroy(st->ev_ready);
    if (st->ev_side) (void)hipEventDestroy(st->ev_side);
    if (st->side) (void)hipStreamDestroy(st->side);
    if (st->drain_sig) (void)hipFree(st->drain_sig);
    for (void* m : st->dl_mem) if (m) (void)hipFree(m);
    for (void* m : st->ser_mem) if (m) (void)hipFree(m);
    if (st->t4_spill_c) (void)hipFree(st->t4_spill_c);
    if (st->t4_spill_a) (void)hipFree(st->t4_spill_a);
    if (st->host_counters) (void)hipHostFree(st->host_counters);
    if (st->ser_host) (void)hipHostFree(st->ser_host);
    delete st;
}
template <class T> static int wf_alloc(WavefrontState* st, T** p, size_t n) {
    WF_TRY(hipMalloc((void**)p, n * sizeof(T)));
    st->mem[st->n_mem++] = (void*)*p;
    return FTN_OK;
}
static int wf_reserve(WavefrontState* st, size_t n) {
    if (n <= st->cap_paths) return FTN_OK;
    wf_free(st);
    WfBuffers& W = st->W; int rc;
    if ((rc = wf_alloc(st, &W.ray, 4 * n)) || (rc = wf_alloc(st, &W.hit, 2 * n)) || (rc = wf_alloc(st, &W.hit_prim, 2 * n)) ||
        (rc = wf_alloc(st, &W.sh, 2 * n)) || (rc = wf_alloc(st, &W.occluded, 2 * n)) || (rc = wf_alloc(st, &W.beta, n)) || (rc = wf_alloc(st, &W.rad, n)) ||
        (rc = wf_alloc(st, &W.rng01, n)) || (rc = wf_alloc(st, &W.rng23, n)) || (rc = wf_alloc(st, &W.pend0, n)) || (rc = wf_alloc(st, &W.pend1, n)) || (rc = wf_alloc(st, &W.pend2, n)) ||
        (rc = wf_alloc(st, &W.q_active[0], n)) || (rc = wf_alloc(st, &W.q_active[1], n)) || (rc = wf_alloc(st, &W.q_closest, 2 * n)) || (rc = wf_alloc(st, &W.q_shadow, 2 * n)) || (rc = wf_alloc(st, &W.q_sorted, 8 * n)) || (rc = wf_alloc(st, &W.cls, 8 * 32)) ||
        (rc = wf_alloc(st, &W.q_exc_closest, 2 * n)) || (rc = wf_alloc(st, &W.q_exc_any, 2 * n)) || (rc = wf_alloc(st, &W.counters, 64 * 32)) || (rc = wf_alloc(st, &st->br, 2 * n)) || (rc = wf_alloc(st, &st->pd, 4 * n))) return rc;
    st->cap_paths = n;
    return FTN_OK;
}

/* tuning knobs (environment overrides, for experiments only): the environment is read once per entry-point call (knobs_begin), not
 * once per launch of the bounce loop */
#include <unordered_map>
static thread_local std::unordered_map<std::string, std::pair<bool, uint32_t>> g_knobs;
static void knobs_begin() { g_knobs.clear(); }
static uint32_t knob(const char* name, uint32_t def) {
    auto it = g_knobs.find(name);
    if (it == g_knobs.end()) { const char* v = getenv(name); it = g_knobs.emplace(name, std::make_pair(v != nullptr, v ? (uint32_t)atoi(v) : 0u)).first; }
    return it->second.first ? it->second.second : def;
}

/* sorts queue[0, cnt) by the rays' coherence keys into `out`; *sorted_q = out (or the queue itself when it is too short to bother) */
static int sort_ray_queue(WavefrontState* st, const RenderParams& P, const WfBuffers& W, bool any, uint32_t* queue, uint32_t cnt, uint32_t* out,
                          uint32_t* k_in, uint32_t* k_out, uint32_t bits, hipStream_t stream, const uint32_t** sorted_q) {
    *sorted_q = queue;
    if (cnt < knob("FTN_WF_SORT_MIN", 16384u)) return FTN_OK;             /* short queues: the launch overhead of the sort outweighs its gain */
    if (any) hipLaunchKernelGGL(k_wf_ray_keys<true>, dim3((cnt + 255) / 256), dim3(256), 0, stream, P.S, W, queue, cnt, k_in, bits);
    else hipLaunchKernelGGL(k_wf_ray_keys<false>, dim3((cnt + 255) / 256), dim3(256), 0, stream, P.S, W, queue, cnt, k_in, bits);
    const int end_bit = (int)(3 * bits + 3);
    size_t need = 0;
    hipcub::DoubleBuffer<uint32_t> keys(k_in, k_out), vals(queue, out);      /* both halves are scratch: the sort ping-pongs instead of copying */
    WF_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys, vals, (int)cnt, 0, end_bit, stream));
    if (need > st->sort_tmp_bytes) { if (st->sort_tmp) (void)hipFree(st->sort_tmp); st->sort_tmp = nullptr; st->sort_tmp_bytes = 0; WF_TRY(hipMalloc(&st->sort_tmp, need)); st->sort_tmp_bytes = need; }
    WF_TRY(hipcub::DeviceRadixSort::SortPairs(st->sort_tmp, need, keys, vals, (int)cnt, 0, end_bit, stream));
    *sorted_q = vals.Current();
    return FTN_OK;
}

/* tuning knobs of k_wf_trace (env overrides are for experiments only) */

#ifdef FTN_DRAIN_PROBE
static uint32_t g_probe_bounce = 0;
#endif
/* Sizes the four-box kernels' launches for this call (LDS levels -> occupancy, persistent grid) and makes sure their spill areas exist.
 * FTN_TRACE4=0 or a scene without four-box records: the two-record kernels run. */
static int trace4_prepare(WavefrontState* st, const DScene& S) {
    st->t4_on = S.quad != nullptr && knob("FTN_TRACE4", 1) != 0;
    if (!st->t4_on) return FTN_OK;
    st->t4 = trace4_plan(S, st->n_cu, knob("FTN_T4_ENTRIES", 0), knob("FTN_T4_ENTRIES_ANY", 0), knob("FTN_T4_WG", 0), knob("FTN_T4_WG_ANY", 0), knob("FTN_T8_WG", 0), knob("FTN_T8_ENTRIES", 0));
    const size_t need_c = (size_t)st->t4.grid_closest * 256u * st->t4.spill_closest * sizeof(uint2);
    /* eight-box occlusion records for the any-hit rays of triangle-only scenes (FTN_T8=0: the four-box kernels trace them) */
    st->t8_on = st->t4.oct_ok && S.n_spheres == 0 && knob("FTN_T8", 1) != 0;
    const size_t need_a8 = st->t8_on ? (size_t)st->t4.grid_oct * 256u * st->t4.spill_oct * sizeof(uint32_t) : 0;
    const size_t need_a = std::max<size_t>(need_a8, (size_t)st->t4.grid_any * 256u * st->t4.spill_any * sizeof(uint32_t));
    if (need_c > st->t4_spill_c_bytes) { if (st->t4_spill_c) (void)hipFree(st->t4_spill_c); st->t4_spill_c = nullptr; st->t4_spill_c_bytes = 0; WF_TRY(hipMalloc(&st->t4_spill_c, need_c)); st->t4_spill_c_bytes = need_c; }
    if (need_a > st->t4_spill_a_bytes) { if (st->t4_spill_a) (void)hipFree(st->t4_spill_a); st->t4_spill_a = nullptr; st->t4_spill_a_bytes = 0; WF_TRY(hipMalloc(&st->t4_spill_a, need_a)); st->t4_spill_a_bytes = need_a; }
    return FTN_OK;
}

/* count: 0 = production kernels, 1 = counting build of the REFERENCE walk (node / primitive tallies equal the oracle's), 2 = counting
 * build of the production kernels (what bench.py's byte model uses).  n_queue: upper bound of the queue's length (sizes the grid). */
static void launch_trace(WavefrontState* st, bool any, int count, bool spheres, unsigned grid, size_t lds, hipStream_t stream, const RenderParams& P, const WfBuffers& W,
                         const uint32_t* queue, const uint32_t* count_ptr, uint32_t* head, uint32_t max_rays, bool camera_rays = false) {
    if (st->t4_on && count != 1) {     /* four-box records */
        const Trace4Plan& T = st->t4;
        const bool oct = any && st->t8_on;
        const unsigned g = std::min<unsigned>(oct ? T.grid_oct : (any ? T.grid_any : T.grid_closest), std::max<unsigned>(1u, (max_rays + 255u) / 256u));
        uint32_t chunk4 = knob("FTN_TRACE_CHUNK", 128);
        while (chunk4 > 64u && (uint64_t)chunk4 * g * 4u * 4u > (uint64_t)max_rays) chunk4 >>= 1;
        /* measured optima on the config-5 scene (profiles/r02_*, r03): lanes re-armed once 32 (closest, four-box any-hit) / 24 (eight-box any-hit)
         * are idle, leaf steps run once 16 lanes hold a leaf, 2 / 4 record steps per control round.
         * The camera rays' launch is the exception: its 64 rays per wave are four pixels' samples and walk the same records, so a wave that
         * re-arms only when ALL of its lanes are done keeps them in lockstep (every load of a step hits the same lines, the ray setup runs once
         * per 64 rays at full width): 46.2 -> 37.1 ms per step with FTN_T4_REFILL0 = 64, leaf steps as soon as 2 lanes hold a leaf, 3 record
         * steps per round (profiles/r03/n_*); at 60 instead of 64 the gain is gone, and the incoherent launches lose 50 % with it.  With
         * the pixels of a tile queued in 2 x 2 blocks (k_wf_generate) a pass of ONE sample per pixel gains too (a wave = an 8 x 8 block:
         * 25.5 -> 24.9 ms at 4096^2; with rows of 16 pixels it lost 2 %) */
        if (oct) launch_trace4(T4K_ANY_OCT, count == 2, false, g, T.lds_oct, T.entries_oct, st->t4_spill_a, stream, P.S, W, queue, count_ptr, head, P.stats,
                               camera_rays ? knob("FTN_T8_REFILL1", 16) : knob("FTN_T8_REFILL", 24), camera_rays ? knob("FTN_T8_LEAF_BATCH1", 16) : knob("FTN_T8_LEAF_BATCH", 16), chunk4, knob("FTN_T8_BURST", 2), knob("FTN_T8_POLICY", 1), T.spill_oct);
        else launch_trace4(any ? T4K_ANY : T4K_CLOSEST, count == 2, spheres, g, any ? T.lds_any : T.lds_closest, any ? T.entries_any : T.entries_closest, any ? st->t4_spill_a : st->t4_spill_c, stream, P.S, W,
                      queue, count_ptr, head, P.stats,
                      any ? knob("FTN_T4_ANY_REFILL", 32) : (camera_rays ? knob("FTN_T4_REFILL0", 64) : knob("FTN_T4_REFILL", 32)),
                      any ? knob("FTN_T4_ANY_LEAF_BATCH", 16) : (camera_rays ? knob("FTN_T4_LEAF_BATCH0", 2) : knob("FTN_T4_LEAF_BATCH", 16)), chunk4,
                      any ? knob("FTN_T4_ANY_BURST", 4) : (camera_rays ? knob("FTN_T4_BURST0", 3) : knob("FTN_T4_BURST", 2)), knob("FTN_T4_ANY_POLICY", 1), any ? T.spill_any : T.spill_closest);
        /* the rays it handed back (a zero direction component: ftn_trace4.hip, point 5) take the reference-order kernel.  The queue is
         * nearly always empty and the persistent workgroups leave at once; its rays are already in the statistics (stats = NULL) */
        const unsigned ge = std::min<unsigned>(grid, (unsigned)st->n_cu);
        uint32_t* ecount = &W.counters[CTR(any ? 33 : 32)]; uint32_t* ehead = &W.counters[CTR(any ? 48 : 40)];
        DevStats* none = nullptr;
#define FTN_TRE(A, Sp) hipLaunchKernelGGL((k_wf_trace<A, false, Sp, 8>), dim3(ge), dim3(256), lds, stream, P.S, W, any ? W.q_exc_any : W.q_exc_closest, ecount, ehead, none, 24u, 2u, 64u, 8u)
        if (any) { if (spheres) FTN_TRE(true, true); else FTN_TRE(true, false); } else { if (spheres) FTN_TRE(false, true); else FTN_TRE(false, false); }
#undef FTN_TRE
        return;
    }
    const bool count_b = count != 0;
    const uint32_t refill = knob("FTN_TRACE_REFILL", 24), leaf_batch = knob("FTN_TRACE_LEAF_BATCH", 2), chunk_knob = knob("FTN_TRACE_CHUNK", 128), node_burst = knob("FTN_TRACE_BURST", 8);
    /* per-wave chunk: large enough that queue-head atomics are rare, small enough that the tail spreads over all waves */
    uint32_t chunk = chunk_knob;
    const uint32_t waves = grid * 4u;
    while (chunk > 64u && (uint64_t)chunk * waves * 4u > (uint64_t)max_rays) chunk >>= 1;
    if (any && !count_b && knob("FTN_TRACE_ANY2", 1) && P.S.fat) {     /* any-hit rays: two boxes per step (k_wf_trace_any2) */
        const uint32_t refill2 = knob("FTN_ANY2_REFILL", 16), leaf_batch2 = knob("FTN_ANY2_LEAF_BATCH", leaf_batch);
        uint32_t policy2 = knob("FTN_ANY2_POLICY", 2);
#ifdef FTN_DRAIN_PROBE
        policy2 |= g_probe_bounce << 8;
#endif
        if (spheres) hipLaunchKernelGGL((k_wf_trace_any2<true>), dim3(grid), dim3(256), lds, stream, P.S, W, queue, count_ptr, head, P.stats, refill2, leaf_batch2, chunk, policy2);
        else hipLaunchKernelGGL((k_wf_trace_any2<false>), dim3(grid), dim3(256), lds, stream, P.S, W, queue, count_ptr, head, P.stats, refill2, leaf_batch2, chunk, policy2);
        return;
    }
    lds += knob("FTN_TRACE_LDS_PAD", 0);     /* experiment: lower the occupancy */
#define FTN_TR(A, C, Sp, B) hipLaunchKernelGGL((k_wf_trace<A, C, Sp, B>), dim3(grid), dim3(256), lds, stream, P.S, W, queue, count_ptr, head, P.stats, refill, leaf_batch, chunk, node_burst)
    const bool fixed = node_burst == 8 && !count_b && !knob("FTN_TRACE_NO_UNROLL", 0);      /* the default burst is compiled in (unrolled) */
    if (fixed) { if (any) { if (spheres) FTN_TR(true, false, true, 8); else FTN_TR(true, false, false, 8); } else { if (spheres) FTN_TR(false, false, true, 8); else FTN_TR(false, false, false, 8); } }
    else if (any) { if (count_b) { if (spheres) FTN_TR(true, true, true, 0); else FTN_TR(true, true, false, 0); } else { if (spheres) FTN_TR(true, false, true, 0); else FTN_TR(true, false, false, 0); } }
    else { if (count_b) { if (spheres) FTN_TR(false, true, true, 0); else FTN_TR(false, true, false, 0); } else { if (spheres) FTN_TR(false, false, true, 0); else FTN_TR(false, false, false, 0); } }
#undef FTN_TR
}

/* buffers of k_wf_shade_dl for n paths, `levels` chain levels and `slots` shadow rays per path; tex: the differentials of textured scenes */
static void wf_free_dl(WavefrontState* st) { for (void*& m : st->dl_mem) { if (m) (void)hipFree(m); m = nullptr; } st->dl_paths = 0; st->dl_levels = 0; st->dl_slots = 0; st->dl_tex = false; }
static int wf_reserve_dl(WavefrontState* st, size_t n, uint32_t levels, uint32_t slots, bool tex) {
    if (n <= st->dl_paths && levels <= st->dl_levels && slots <= st->dl_slots && (!tex || st->dl_tex)) return FTN_OK;
    wf_free_dl(st);
    const size_t sl = std::max<uint32_t>(slots, 2u);                       /* (direct lighting: shadow results at [0, n), MIS-any results at [n, 2n)) */
    WF_TRY(hipMalloc(&st->dl_mem[0], (size_t)levels * n * sizeof(float4)));          /* dlA */
    WF_TRY(hipMalloc(&st->dl_mem[1], (size_t)levels * n * sizeof(float4)));          /* dlB */
    WF_TRY(hipMalloc(&st->dl_mem[2], (size_t)slots * n * sizeof(float4)));           /* whT */
    WF_TRY(hipMalloc(&st->dl_mem[3], 2 * (size_t)slots * n * sizeof(float4)));       /* shadow-ray records */
    WF_TRY(hipMalloc(&st->dl_mem[4], sl * n));                                       /* occluded */
    WF_TRY(hipMalloc(&st->dl_mem[5], sl * n * sizeof(uint32_t)));                    /* any-hit queue */
    WF_TRY(hipMalloc(&st->dl_mem[6], sl * n * sizeof(uint32_t)));                    /* ... and the queue of the rays the four-box kernel hands back */
    if (tex) WF_TRY(hipMalloc(&st->dl_mem[7], 3 * n * sizeof(float4)));              /* dfd */
    st->dl_paths = n; st->dl_levels = levels; st->dl_slots = slots; st->dl_tex = tex;
    return FTN_OK;
}

static int wf_state_init(WavefrontState** state) {
    if (*state) return FTN_OK;
    *state = new WavefrontState();
    for (int i = 0; i < 64; i++) { WF_TRY(hipEventCreate(&(*state)->ev[i])); (*state)->n_ev = i + 1; }
    {   /* lowest priority: its workgroups should only take what the main stream's kernels leave free */
        int least = 0, greatest = 0;
        WF_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        WF_TRY(hipStreamCreateWithPriority(&(*state)->side, hipStreamNonBlocking, knob("FTN_WF_SIDE_PRIO", 1) ? least : 0));
    }
    WF_TRY(hipEventCreateWithFlags(&(*state)->ev_ready, hipEventDisableTiming));
    WF_TRY(hipEventCreateWithFlags(&(*state)->ev_side, hipEventDisableTiming));
    {   /* stream memory operations: the side stream can wait for a value a running kernel stores */
        int dev = 0, can = 0;
        WF_TRY(hipGetDevice(&dev));
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) == hipSuccess && can) {
            if (hipExtMallocWithFlags((void**)&(*state)->drain_sig, 8, hipMallocSignalMemory) != hipSuccess) { (void)hipGetLastError(); (*state)->drain_sig = nullptr; }
            else WF_TRY(hipMemset((*state)->drain_sig, 0, 8));
        }
    }
    WF_TRY(hipHostMalloc((void**)&(*state)->host_counters, 16 * 32 * sizeof(uint32_t)));
    hipDeviceProp_t prop; int dev = 0; WF_TRY(hipGetDevice(&dev)); WF_TRY(hipGetDeviceProperties(&prop, dev));
    (*state)->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return FTN_OK;
}

/* ------------------------------------------------------------------ Scene::intersect / intersect_test for arrays of rays (ftn_intersect*):
 * the same traversal kernel (and coherence sort) the renderer uses, over a queue that simply lists the rays */
__global__ void __launch_bounds__(256) k_batch_setup(const float* __restrict__ rays8, uint32_t n, float4* __restrict__ ray, uint32_t* __restrict__ queue) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float* r = rays8 + 8 * (size_t)i;
    ray[2 * (size_t)i] = make_float4(r[0], r[1], r[2], r[7]);            /* o, time */
    ray[2 * (size_t)i + 1] = make_float4(r[3], r[4], r[5], r[6]);        /* d, t_max */
    queue[i] = i;
}
__global__ void __launch_bounds__(256) k_batch_finish(DScene S, const float* __restrict__ rays8, uint32_t n, const float4* __restrict__ hit, const int* __restrict__ hit_prim,
                                                      float* t_hit, int* prim, float* bary, float* out24) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 h4 = hit[i]; const int hp = hit_prim[i];
    if (t_hit) t_hit[i] = hp >= 0 ? h4.x : FTN_INF;
    if (prim) prim[i] = hp;
    if (bary) { bary[3 * (size_t)i] = hp >= 0 ? h4.y : 0.0f; bary[3 * (size_t)i + 1] = hp >= 0 ? h4.z : 0.0f; bary[3 * (size_t)i + 2] = hp >= 0 ? h4.w : 0.0f; }
    if (out24) {                                                        /* the full SurfaceInteraction of the hit (ftn_intersect_full) */
        float* o = out24 + 24 * (size_t)i;
        if (hp < 0) { for (int k = 0; k < 24; k++) o[k] = 0.0f; o[23] = -1.0f; }
        else {
            const float* r = rays8 + 8 * (size_t)i;
            DRay ray0; ray0.o = V3(r[0], r[1], r[2]); ray0.d = V3(r[3], r[4], r[5]); ray0.t_max = r[6]; ray0.time = r[7];
            DHit h; h.t = h4.x; h.b0 = h4.y; h.b1 = h4.z; h.b2 = h4.w; h.prim = hp;
            DSI si; make_interaction(S, h, ray0, &si);
            o[0] = si.hit.p.x; o[1] = si.hit.p.y; o[2] = si.hit.p.z; o[3] = si.hit.p_err.x; o[4] = si.hit.p_err.y; o[5] = si.hit.p_err.z;
            o[6] = si.hit.n.x; o[7] = si.hit.n.y; o[8] = si.hit.n.z; o[9] = 0.0f; o[10] = 0.0f;
            o[11] = si.wo.x; o[12] = si.wo.y; o[13] = si.wo.z; o[14] = si.s_dpdu.x; o[15] = si.s_dpdu.y; o[16] = si.s_dpdu.z;
            o[17] = 0.0f; o[18] = 0.0f; o[19] = 0.0f; o[20] = si.shading_n.x; o[21] = si.shading_n.y; o[22] = si.shading_n.z; o[23] = h.t;
        }
    }
}
int wavefront_trace_batch(WavefrontState** state, const DScene& S, uint32_t stack_entries, const float* d_rays8, size_t n_rays, int mode, bool count,
                          float* t_hit, int* prim, float* bary, unsigned char* occluded, float* out24, DevStats* stats, hipStream_t stream) {
    if (n_rays == 0) return FTN_OK;
    if (n_rays >= (1ull << 31)) { g_wf_err = "too many rays in one batch"; return FTN_ERR_INVALID_ARGUMENT; }
    knobs_begin();
    int rc = wf_state_init(state); if (rc) return rc;
    WavefrontState* st = *state;
    if ((rc = trace4_prepare(st, S))) return rc;
    const uint32_t n = (uint32_t)n_rays;
    const bool any = mode == 1;
    float4 *ray = nullptr, *hit = nullptr; int* hit_prim = nullptr; uint32_t *queue = nullptr, *scratch = nullptr, *counters = nullptr, *q_exc = nullptr;
    auto cleanup = [&]() { (void)hipFree(ray); (void)hipFree(hit); (void)hipFree(hit_prim); (void)hipFree(queue); (void)hipFree(scratch); (void)hipFree(counters); (void)hipFree(q_exc); };
#define WFB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_wf_err = std::string(#expr ": ") + hipGetErrorString(e_); cleanup(); return e_ == hipErrorOutOfMemory ? FTN_ERR_OUT_OF_MEMORY : FTN_ERR_NO_DEVICE; } } while (0)
    WFB_TRY(hipMalloc((void**)&ray, (size_t)n * 32)); WFB_TRY(hipMalloc((void**)&queue, (size_t)n * 4)); WFB_TRY(hipMalloc((void**)&scratch, (size_t)n * 12));
    WFB_TRY(hipMalloc((void**)&counters, 64 * 32 * 4)); WFB_TRY(hipMemsetAsync(counters, 0, 64 * 32 * 4, stream));
    WFB_TRY(hipMalloc((void**)&q_exc, (size_t)n * 4));
    if (!any) { WFB_TRY(hipMalloc((void**)&hit, (size_t)n * 16)); WFB_TRY(hipMalloc((void**)&hit_prim, (size_t)n * 4)); }
    hipLaunchKernelGGL(k_batch_setup, dim3((n + 255) / 256), dim3(256), 0, stream, d_rays8, n, ray, queue);
    WFB_TRY(hipMemcpyAsync(&counters[CTR(any ? 3 : 2)], &n, 4, hipMemcpyHostToDevice, stream));
    WfBuffers W; memset(&W, 0, sizeof(W));
    W.n_paths = n; W.ray = ray; W.sh = ray; W.hit = hit; W.hit_prim = hit_prim; W.occluded = occluded; W.counters = counters; W.q_exc_closest = q_exc; W.q_exc_any = q_exc;
    RenderParams P; memset(&P, 0, sizeof(P)); P.S = S; P.stats = stats;
    const uint32_t* q = queue;
    if (knob("FTN_WF_SORT", 0)) { rc = sort_ray_queue(st, P, W, any, queue, n, scratch, scratch + n, scratch + 2 * (size_t)n, 7, stream, &q); if (rc) { cleanup(); return rc; } }
    const size_t lds = (size_t)stack_entries * 256 * sizeof(uint32_t);
    const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(8, (size_t)(160 * 1024) / std::max<size_t>(lds, 1)));
    const unsigned grid = std::min<unsigned>((unsigned)st->n_cu * per_cu, (n + 255) / 256);
    launch_trace(st, any, count ? 1 : 0, S.n_spheres != 0, grid, lds, stream, P, W, q, &counters[CTR(any ? 3 : 2)], &counters[CTR(any ? 24 : 16)], n);
    if (!any) hipLaunchKernelGGL(k_batch_finish, dim3((n + 255) / 256), dim3(256), 0, stream, S, d_rays8, n, hit, hit_prim, t_hit, prim, bary, out24);
    WFB_TRY(hipStreamSynchronize(stream));
    WFB_TRY(hipGetLastError());
#undef WFB_TRY
    cleanup();
    return FTN_OK;
}

/* ------------------------------------------------------------------ FTN_SAMPLER_TILE_SERIAL through the queues (PathIntegrator).
 * The reference's RandomSampler is one serial stream per tile, so a tile has ONE camera sample in flight; with T tiles the wavefront holds
 * T paths at mixed depths.  A round = the two traces, classify, the shade launches (the kernels of the indexed pipeline, unchanged) and
 * k_wf_serial_advance, which retires finished paths into the film and starts the tiles' next samples; the host polls the active count
 * once per round.  Same stream, same film sums as k_render_serial -- but the traversal and the shading of a round run convergent instead
 * of diverging inside one megakernel lane per tile.  Measured on the config-5 scene (65,536 tiles, 1 spp): 1110 rounds of 0.51 ms = 94 Mrays/s
 * against the megakernel's 35.  A round is as long as its longest walk (latency, not throughput: the GPU holds 5 x these rays in flight),
 * so cutting the tiles into groups that run side by side multiplies the rounds without shortening any (profiles/r03/d_*: not kept). */
static int wavefront_render_serial(WavefrontState* st, const RenderParams& P, uint32_t n_tiles, bool count, bool count_production, hipStream_t stream, WavefrontTimes* times) {
    int rc = wf_reserve(st, (size_t)std::max<uint32_t>(n_tiles, 256u));
    if (rc) { wf_free(st); return rc; }
    const bool tex = P.S.n_textures != 0;
    if (n_tiles > st->ser_paths || (tex && !st->ser_tex)) {
        for (void*& m : st->ser_mem) { if (m) (void)hipFree(m); m = nullptr; }
        st->ser_paths = 0;
        WF_TRY(hipMalloc(&st->ser_mem[0], (size_t)n_tiles * sizeof(uint2)));
        WF_TRY(hipMalloc(&st->ser_mem[1], (size_t)n_tiles * sizeof(float2)));
        WF_TRY(hipMalloc(&st->ser_mem[2], (size_t)n_tiles));
        if (tex) WF_TRY(hipMalloc(&st->ser_mem[3], 3 * (size_t)n_tiles * sizeof(float4)));
        st->ser_paths = n_tiles; st->ser_tex = tex;
    }
    if ((rc = trace4_prepare(st, P.S))) return rc;
    WfBuffers W = st->W;
    W.serial = 1; W.ser_cursor = (uint2*)st->ser_mem[0]; W.ser_pfilm = (float2*)st->ser_mem[1]; W.ser_retired = (unsigned char*)st->ser_mem[2]; W.dfd = (float4*)st->ser_mem[3];
    W.n_slots = 0; W.samples = 1; W.n_paths = n_tiles; W.first_sample = 0; W.seg_cap = (uint32_t)st->cap_paths; W.valid_per_sample = 0;
    W.gen_blocks = 0; W.rng_replay = 0; W.br = nullptr; W.pd = nullptr; W.pd_md = 0; W.pd_occ = 0;
    const int count_mode = count ? (count_production ? 2 : 1) : 0;
    W.mis_any = ((!count || count_production) && knob("FTN_MIS_ANY", 1)) ? 1u : 0u;
    const bool spheres = P.S.n_spheres != 0;
    const size_t lds = (size_t)P.stack_entries * 256 * sizeof(uint32_t);
    const unsigned blocks_per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(8, (size_t)(160 * 1024) / std::max<size_t>(lds, 1)));
    const unsigned tg = std::min<unsigned>((unsigned)st->n_cu * blocks_per_cu, (2 * n_tiles + 255) / 256);
    const dim3 pgrid((n_tiles + 255) / 256), sgrid(std::min<unsigned>((unsigned)st->n_cu * 8u, (n_tiles + 255) / 256));
    const bool env = P.S.env_only && knob("FTN_SHADE_ENV", 1);
    hipEvent_t e0 = st->ev[0], e1 = st->ev[1];
    (void)hipEventRecord(e0, stream);
    hipLaunchKernelGGL(k_wf_reset, dim3(1), dim3(64), 0, stream, W, 3, 0, P.stats);
    if (tex) hipLaunchKernelGGL((k_wf_serial_advance<true>), pgrid, dim3(256), 0, stream, P, W, 0, 1); else hipLaunchKernelGGL((k_wf_serial_advance<false>), pgrid, dim3(256), 0, stream, P, W, 0, 1);
    int in_q = 0; unsigned long long rounds = 0, trace_launches = 0;
    if (!st->ser_host) WF_TRY(hipHostMalloc((void**)&st->ser_host, 2 * 16 * 32 * sizeof(uint32_t)));
    int polled_q[2] = {0, 0}; uint32_t* last = st->ser_host;
    const bool side_ok = st->side != nullptr && knob("FTN_WF_OVERLAP", 2) != 0 && getenv("ROCPROF_COUNTER_COLLECTION") == nullptr;
    /* every round ends at least one bounce of every live tile; a sample takes at most max_depth + 3 rounds, plus null-material pass-throughs */
    const uint64_t max_rounds = 256u * (uint64_t)std::max<uint32_t>(P.spp, 1u) * ((uint64_t)P.max_depth + 4u + (1u << 12));
    for (uint64_t it = 0; it < max_rounds; it++) {
        /* the two traces of a round side by side: with one ray per tile a launch is as long as its longest walk (0.25 - 0.35 ms for 65 k
         * rays that the GPU could trace in 0.02 ms), and the two launches are independent */
        const bool beside = it > 0 && side_ok;
        if (beside) { WF_TRY(hipEventRecord(st->ev_ready, stream)); WF_TRY(hipStreamWaitEvent(st->side, st->ev_ready, 0));
                      launch_trace(st, true, count_mode, spheres, tg, lds, st->side, P, W, W.q_shadow, &W.counters[CTR(3)], &W.counters[CTR(24)], 2 * n_tiles);
                      WF_TRY(hipEventRecord(st->ev_side, st->side)); }
        launch_trace(st, false, count_mode, spheres, tg, lds, stream, P, W, W.q_closest, &W.counters[CTR(2)], &W.counters[CTR(16)], 2 * n_tiles);
        trace_launches++;
        if (it > 0 && !beside) launch_trace(st, true, count_mode, spheres, tg, lds, stream, P, W, W.q_shadow, &W.counters[CTR(3)], &W.counters[CTR(24)], 2 * n_tiles);
        if (beside) WF_TRY(hipStreamWaitEvent(stream, st->ev_side, 0));
        hipLaunchKernelGGL(k_wf_reset, dim3(1), dim3(64), 0, stream, W, 2, in_q, P.stats);
        hipLaunchKernelGGL(k_wf_classify, sgrid, dim3(256), 0, stream, P, W, in_q);
        hipLaunchKernelGGL(k_wf_reset, dim3(1), dim3(64), 0, stream, W, 1, in_q, P.stats);
#define FTN_SH2(T, M, E, mask) hipLaunchKernelGGL((k_wf_shade<T, M, E>), sgrid, dim3(256), 0, stream, P, W, in_q, mask, 0u)
#define FTN_SH(M, mask) do { if (tex) { if (env) FTN_SH2(true, M, true, mask); else FTN_SH2(true, M, false, mask); } \
                             else { if (env) FTN_SH2(false, M, true, mask); else FTN_SH2(false, M, false, mask); } } while (0)
        if (!knob("FTN_SHADE_SPECIALISE", 1)) FTN_SH(-1, 0xffu);
        else {
            FTN_SH(-2, 0x83u);
            if (P.S.material_types & 1u) FTN_SH(0, 1u << 2);
            if (P.S.material_types & 2u) FTN_SH(1, 1u << 3);
            if (P.S.material_types & 4u) FTN_SH(2, 1u << 4);
            if (P.S.material_types & 8u) FTN_SH(3, 1u << 5);
            if (P.S.material_types & 16u) FTN_SH(4, 1u << 6);
        }
#undef FTN_SH2
#undef FTN_SH
        if (tex) hipLaunchKernelGGL((k_wf_serial_advance<true>), pgrid, dim3(256), 0, stream, P, W, in_q ^ 1, 0); else hipLaunchKernelGGL((k_wf_serial_advance<false>), pgrid, dim3(256), 0, stream, P, W, in_q ^ 1, 0);
        in_q ^= 1; rounds++;
        /* the active count of this round is read while the NEXT round is already queued (no bubble on the GPU between rounds); the loop
         * therefore ends one round late, with a round over empty queues */
        uint32_t* hb = st->ser_host + (it & 1u) * 16u * 32u;
        WF_TRY(hipMemcpyAsync(hb, W.counters, 16 * 32 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        WF_TRY(hipEventRecord(st->ev[2 + (it & 1u)], stream));
        polled_q[it & 1u] = in_q;
        if (it >= 1) {
            const uint32_t k = (uint32_t)((it - 1) & 1u);
            WF_TRY(hipEventSynchronize(st->ev[2 + k]));
            if (st->ser_host[k * 16u * 32u + CTR(polled_q[k] == 0 ? 0 : 1)] == 0) { last = hb; break; }
        }
        last = hb;
    }
    WF_TRY(hipStreamSynchronize(stream));
    memcpy(st->host_counters, last, 16 * 32 * sizeof(uint32_t));
    (void)hipEventRecord(e1, stream);
    WF_TRY(hipEventSynchronize(e1));
    WF_TRY(hipGetLastError());
    if (times) {
        float ms = 0.0f; (void)hipEventElapsedTime(&ms, e0, e1);
        memset(times, 0, sizeof(*times));
        times->trace_launches = trace_launches; times->mis_any_rays = W.mis_any ? st->host_counters[CTR(10)] : 0;
        times->shade_launches = rounds;
        (void)ms;
    }
    return FTN_OK;
}

int wavefront_render(WavefrontState** state, const RenderParams& P0, const std::vector<DTile>& tiles, bool count, hipStream_t stream, WavefrontTimes* times,
                     bool count_production) {
    knobs_begin();
    { int rc0 = wf_state_init(state); if (rc0) return rc0; }
    WavefrontState* st = *state;
    RenderParams P = P0;
    const uint32_t n_slots = (uint32_t)tiles.size() * 256u;
    if (n_slots == 0) return FTN_OK;
    const uint32_t total_samples = P.last_sample - P.first_sample;
    if (total_samples == 0) return FTN_OK;
    if (tiles.size() > ((size_t)1 << 20)) { g_wf_err = "more than 2^20 tiles (2^28 pixel slots) in one call: render the film in several tile ranges"; return FTN_ERR_UNSUPPORTED; }
    if (P.sampler_kind != FTN_SAMPLER_INDEXED) {
        if (P.integrator_kind != FTN_INTEGRATOR_PATH) { g_wf_err = "the tile-serial sampler runs on the queues for PathIntegrator"; return FTN_ERR_UNSUPPORTED; }
        return wavefront_render_serial(st, P, (uint32_t)tiles.size(), count, count_production, stream, times);
    }
    /* samples per pass: up to 256 Mi paths in flight (~88 GB of path state and queues out of 288 GB; the ids, queue indices and sort
     * counts are 32-bit: 2^28 paths is also their limit).  Bigger wavefronts are faster per ray: the sorted queues hold more rays per
     * cell of space (more lanes of a wave share node records) and every launch's drain -- about 0.5 ms whatever its size -- is paid
     * once for more work.  Measured on the config-5 scene, per sample per pixel: 16 Mi paths 29.7 ms, 32 Mi 27.8, 64 Mi 26.7,
     * 128 Mi 25.1, 256 Mi 24.5 (FTN_WF_PATHS_M, in Mi paths).  If the GPU has less to give, the pass is halved (below). */
    uint32_t S = (uint32_t)std::max<size_t>(1, ((size_t)std::min<uint32_t>(knob("FTN_WF_PATHS_M", 256), 256u) << 20) / n_slots);
    S = std::min(S, total_samples);
    const int count_mode = count ? (count_production ? 2 : 1) : 0;
    const bool dl_mode = P.integrator_kind != FTN_INTEGRATOR_PATH;          /* DirectLightingIntegrator / WhittedIntegrator: k_wf_shade_dl */
    const bool whitted = P.integrator_kind == FTN_INTEGRATOR_WHITTED;
    const bool dl_tex = dl_mode && P.S.n_textures != 0;
    uint32_t dl_levels = 0, dl_slots = 0;
    if (dl_mode) {
        if (whitted && P.S.n_lights > WF_WH_MAX_LIGHTS) { g_wf_err = "the wavefront pipeline runs WhittedIntegrator for up to 32 lights (one bit per light in a path's pending-light word)"; return FTN_ERR_UNSUPPORTED; }
        dl_levels = std::max<uint32_t>(1u, std::min<uint32_t>(P.max_depth, WF_DL_MAX)); dl_slots = whitted ? std::max<uint32_t>(P.S.n_lights, 1u) : 1u;
        /* a shadow ray's queue entry is its slot p + l * n_paths, and bit 31 marks MIS rays: slots * paths stays below 2^31 */
        const size_t cap = ((size_t)1 << 31) / std::max<uint32_t>(dl_slots, 2u) - 1u;
        S = (uint32_t)std::max<size_t>(1, std::min<size_t>(S, cap / n_slots));
        if ((size_t)n_slots > cap) { g_wf_err = "too many pixel slots for this many lights in one call: render the film in several tile ranges"; return FTN_ERR_UNSUPPORTED; }
    }
    auto reserve = [&](uint32_t samples) -> int {
        int r = wf_reserve(st, (size_t)samples * n_slots);
        if (!r && dl_mode) r = wf_reserve_dl(st, st->cap_paths, dl_levels, dl_slots, dl_tex);
        return r;
    };
    int rc = reserve(S);
    while (rc == FTN_ERR_OUT_OF_MEMORY && S > 1) {            /* the wavefront does not fit next to what else lives on this GPU: smaller passes */
        wf_free(st); wf_free_dl(st); (void)hipGetLastError();
        S = (S + 1) / 2;
        rc = reserve(S);
    }
    if (rc) { wf_free(st); wf_free_dl(st); return rc; }
    if ((rc = trace4_prepare(st, P.S))) return rc;
    if (dl_mode) { st->W.dlA = (float4*)st->dl_mem[0]; st->W.dlB = (float4*)st->dl_mem[1]; st->W.whT = (float4*)st->dl_mem[2]; st->W.dfd = (float4*)st->dl_mem[7]; }
    WfBuffers W = st->W;
    if (dl_mode) { W.sh = (float4*)st->dl_mem[3]; W.occluded = (unsigned char*)st->dl_mem[4]; W.q_shadow = (uint32_t*)st->dl_mem[5]; W.q_exc_any = (uint32_t*)st->dl_mem[6]; }
    if (!knob("FTN_ENV_CELLS_USE", 1)) P.S.env0.cells = nullptr;      /* (A/B in one process: the environment-only kernels read the light from the kernel arguments) */
    const bool spheres = P.S.n_spheres != 0;
    uint32_t valid = 0; for (const DTile& t : tiles) valid += (uint32_t)((t.x1 - t.x0) * (t.y1 - t.y0));
    W.valid_per_sample = valid;
    W.mis_any = ((!count || count_production) && knob("FTN_MIS_ANY", 1)) ? 1u : 0u;
    /* the path integrator's streams replayed from the sample key instead of carried (WfBuffers::rng_replay): when the draw count fits its 9 bits */
    W.rng_replay = (!dl_mode && 5u + 8u * ((uint32_t)P.max_depth + 1u) <= 511u && knob("FTN_RNG_REPLAY", 1)) ? 1u : 0u;      /* shading 101.7 -> 96.9 ms per step */
    W.br = (!dl_mode && knob("FTN_WF_BR", 1)) ? st->br : nullptr;
    W.pd = (!dl_mode && knob("FTN_WF_PD", 3)) ? st->pd : nullptr;      /* 1: the three pending terms in one record; 2: + the MIS ray's direction; 3: + the any-hit results (WfBuffers::pd_md, pd_occ) */
    W.pd_md = (W.pd && knob("FTN_WF_PD", 3) >= 2) ? 1u : 0u;
    W.pd_occ = (W.pd && knob("FTN_WF_PD", 3) >= 3) ? 1u : 0u;
    W.gen_blocks = knob("FTN_GEN_BLOCKS", 1);      /* camera rays of a full tile queued in 2 x 2 pixel blocks: first closest-hit launch 37.3 -> 36.6 ms */
    const size_t lds = (size_t)P.stack_entries * 256 * sizeof(uint32_t);
    const unsigned blocks_per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(8, (size_t)(160 * 1024) / std::max<size_t>(lds, 1)));
    const unsigned trace_grid_max = (unsigned)st->n_cu * blocks_per_cu;
    const unsigned shade_grid_max = (unsigned)st->n_cu * 8u;
    double trace_ms = 0.0; unsigned long long trace_launches = 0, mis_any_rays = 0;
    double group_ms[4] = {0.0, 0.0, 0.0, 0.0}; unsigned long long group_n[4] = {0, 0, 0, 0};     /* 0 closest-hit trace, 1 any-hit trace, 2 classify + shade, 3 queue sort */
    /* (rocprofv3's counter collection runs one dispatch at a time and never gets to the launch a stream-memory wait is waiting for:
     * with ROCPROF_COUNTER_COLLECTION set the gate is left out -- kernels are serialised under that tool anyway) */
    /* FTN_WF_OVERLAP: 0 = one stream, 1 = any-hit launches beside the closest-hit ones, 2 (default) = only for wavefronts of up to 32 Mi
     * paths: short launches are mostly drain and gain 4 %, at 128 Mi paths the gain is 0.7 % and not worth the second stream */
    const uint32_t overlap_mode = knob("FTN_WF_OVERLAP", 2);
    const bool drain_gate = knob("FTN_WF_DRAIN_GATE", 1) != 0 && getenv("ROCPROF_COUNTER_COLLECTION") == nullptr;
    /* Coherence sort of the secondary closest-hit queue (FTN_WF_SORT=1): with the two-record kernels of round 1 it paid 6 %; the four-box
     * kernels are instruction-issue bound and gain as much as the sort (and its per-bounce read-back of the queue lengths) costs: 354.1 vs
     * 353.4 ms per step on the config-5 scene, and it loses 6 % on the smaller configurations (profiles/r02).  Off by default. */
    const uint32_t sort_bits = knob("FTN_WF_SORT", 0) ? std::min<uint32_t>(std::max<uint32_t>(knob("FTN_WF_SORT_BITS", 7), 1u), 9u) : 0u;
    uint32_t n_active = 0;                 /* length of the active queue the next classify reads (known from the previous bounce's poll) */
    int ev_used = 0;
    struct Span { int a, b, kind; };
    std::vector<Span> spans;
    static const char* const kGroup[4] = {"closest-hit trace", "any-hit trace", "classify + shade", "queue sort"};
    auto flush_events = [&]() -> int {
        if (spans.empty()) return FTN_OK;
        for (const Span& s : spans) WF_TRY(hipEventSynchronize(st->ev[s.b]));       /* (the any-hit launches may sit on the side stream) */
        for (const Span& s : spans) {
            float ms = 0.0f; (void)hipEventElapsedTime(&ms, st->ev[s.a], st->ev[s.b]);
            group_ms[s.kind] += ms; group_n[s.kind]++;
            if (s.kind == 0) trace_ms += ms;
            if (knob("FTN_WF_DEBUG", 0)) fprintf(stderr, "[wf] %s: %.3f ms\n", kGroup[s.kind], ms);
        }
        spans.clear(); ev_used = 0;
        return FTN_OK;
    };
    /* HIP events around a group of launches on the stream they are launched on */
    auto span_begin = [&](hipStream_t sm) -> int { const int a = ev_used++; (void)hipEventRecord(st->ev[a], sm); return a; };
    auto span_end = [&](int a, int kind, hipStream_t sm) { const int b = ev_used++; (void)hipEventRecord(st->ev[b], sm); spans.push_back(Span{a, b, kind}); };
    /* (a lambda: whatever way it is left -- an error in the middle of a bounce included -- the recorded spans are collected below) */
    auto render_passes = [&]() -> int {
    for (uint32_t s0 = 0; s0 < total_samples; s0 += S) {
        const uint32_t Sp = std::min(S, total_samples - s0);
        W.n_slots = n_slots; W.samples = Sp; W.n_paths = Sp * n_slots; W.first_sample = P.first_sample + s0; W.seg_cap = (uint32_t)st->cap_paths;
        hipLaunchKernelGGL(k_wf_reset, dim3(1), dim3(64), 0, stream, W, 0, 0, P.stats);
        hipLaunchKernelGGL(k_wf_generate, dim3((W.n_paths + 255) / 256), dim3(256), 0, stream, P, W, dl_mode ? 1 : 0);
        int in_q = 0;
        const uint32_t* q_cl = W.q_closest; const uint32_t* q_sh = W.q_shadow;      /* the camera rays' queue is in pixel order already */
        /* null-material pass-throughs do not count as bounces (path.rs:77-81), so the number of rounds has no bound in max_depth alone: the
         * loop ends when a poll finds no active path; the cap only guards against a path that never terminates */
        const uint32_t max_iter = P.max_depth + 2 + (1u << 20);
        for (uint32_t it = 0; it < max_iter; it++) {
            bool polled = false;
            const unsigned tg = std::min<unsigned>(trace_grid_max, (2 * W.n_paths + 255) / 256);
            if (ev_used + 10 > 64) { rc = flush_events(); if (rc) return rc; }
            /* The two traces of a bounce are independent (own queues, own result arrays).  A persistent traversal kernel ends with a
             * long drain -- the last rays are sequential walks of several hundred nodes while most waves have already left (measured:
             * 0.3-0.8 ms from the first idle wave to the end of every launch) -- so the any-hit launch goes to a second stream and its
             * workgroups move in as the closest-hit ones leave.  The closest-hit launch is issued first and still has the GPU to itself
             * until it starts draining, which keeps its event timing (roofline) meaningful. */
#ifdef FTN_DRAIN_PROBE
            g_probe_bounce = it;
#endif
            const bool overlap = overlap_mode == 1u || (overlap_mode == 2u && W.n_paths <= (32u << 20));
            const bool beside = overlap && it > 0 && q_sh == W.q_shadow;      /* (a sorted any-hit queue lives in scratch that classify reuses) */
            W.drain_sig = nullptr; W.drain_seq = 0;
            const bool gated = beside && st->drain_sig && drain_gate;
            if (beside) WF_TRY(hipEventRecord(st->ev_ready, stream));
            if (gated) {
                /* ... and not before the closest-hit launch has started to drain: its first wave that finds the queue dry stores the
                 * launch's sequence number; the write behind the launch releases the waiter in any case */
                if (st->drain_seq >= 0xfffffff0u) { WF_TRY(hipStreamSynchronize(stream)); WF_TRY(hipStreamSynchronize(st->side)); WF_TRY(hipMemset(st->drain_sig, 0, 8)); st->drain_seq = 0; }
                W.drain_sig = st->drain_sig; W.drain_seq = ++st->drain_seq;
                W.drain_at = 1;           /* the first wave (waiting for 10-75 % of the waves to be dry measured the same) */
            }
            { const int e = span_begin(stream);
              launch_trace(st, false, count_mode, spheres, tg, lds, stream, P, W, q_cl, &W.counters[CTR(2)], &W.counters[CTR(16)], 2 * W.n_paths,
                           it == 0 && W.samples >= knob("FTN_T4_LOCKSTEP_MIN_SPP", 1) /* a wave's 64 rays are at most 16 pixels' samples: the lockstep parameters (launch_trace) */);
              span_end(e, 0, stream); }
            if (gated) WF_TRY(hipStreamWriteValue32(stream, st->drain_sig, W.drain_seq, 0));
            trace_launches++;
            /* the side stream's waits are enqueued AFTER the launch they wait for: a tool that executes everything one command at a time
             * in submission order (rocprofv3 --pmc) then meets launch, release, wait -- not a wait nothing can release */
            if (beside) WF_TRY(hipStreamWaitEvent(st->side, st->ev_ready, 0));
            if (gated) WF_TRY(hipStreamWaitValue32(st->side, st->drain_sig, W.drain_seq, hipStreamWaitValueGte, 0xffffffffu));
            if (it > 0) {
                const unsigned sg = std::min<unsigned>(trace_grid_max, (2 * W.n_paths + 255) / 256);
                hipStream_t as = beside ? st->side : stream;
                const int e = span_begin(as);
                launch_trace(st, true, count_mode, spheres, sg, lds, as, P, W, q_sh, &W.counters[CTR(3)], &W.counters[CTR(24)], 2 * W.n_paths, it == 1 && W.samples >= 4u);
                span_end(e, 1, as);
                if (beside) WF_TRY(hipEventRecord(st->ev_side, st->side));
            }
            {   /* group the active paths by shading class (reads the hit records the traces just wrote) */
                const int e_cls = span_begin(stream);
                const unsigned cg = std::min<unsigned>(shade_grid_max, (W.n_paths + 255) / 256);
                hipLaunchKernelGGL(k_wf_reset, dim3(1), dim3(64), 0, stream, W, 2, in_q, P.stats);
                hipLaunchKernelGGL(k_wf_classify, dim3(cg), dim3(256), 0, stream, P, W, in_q);
                span_end(e_cls, 2, stream);
            }
            /* shading needs the occlusion results (classify above did not).  The shade span starts BEHIND this wait: with the any-hit launch
             * on the side stream the wait is any-hit time, which has its own span */
            if (beside) WF_TRY(hipStreamWaitEvent(stream, st->ev_side, 0));
            const int e_shade = span_begin(stream);
            hipLaunchKernelGGL(k_wf_reset, dim3(1), dim3(64), 0, stream, W, 1, in_q, P.stats);
            {
                const dim3 sgrid(std::min<unsigned>(shade_grid_max, (W.n_paths + 255) / 256));
                const bool tex = P.S.n_textures != 0;
                const uint32_t first = it == 0 ? 1u : 0u;
                if (dl_mode) {
                    if (whitted) { if (tex) hipLaunchKernelGGL((k_wf_shade_dl<true, true>), sgrid, dim3(256), 0, stream, P, W, in_q); else hipLaunchKernelGGL((k_wf_shade_dl<true, false>), sgrid, dim3(256), 0, stream, P, W, in_q); }
                    else { if (tex) hipLaunchKernelGGL((k_wf_shade_dl<false, true>), sgrid, dim3(256), 0, stream, P, W, in_q); else hipLaunchKernelGGL((k_wf_shade_dl<false, false>), sgrid, dim3(256), 0, stream, P, W, in_q); }
                } else if (!knob("FTN_SHADE_SPECIALISE", 1)) {
                    if (tex) hipLaunchKernelGGL((k_wf_shade<true, -1, false>), sgrid, dim3(256), 0, stream, P, W, in_q, 0xffu, first);
                    else hipLaunchKernelGGL((k_wf_shade<false, -1, false>), sgrid, dim3(256), 0, stream, P, W, in_q, 0xffu, first);
                } else {   /* one launch per material type present in the scene + one for the classes without a BSDF */
                    const bool env = P.S.env_only && knob("FTN_SHADE_ENV", 1);        /* lit by one InfiniteAreaLight: the variants specialised for it */
                    /* workgroups per CU: what the kernel's registers let a CU hold at once (x 2 for the light kernel of the classes without a BSDF) -- every
                     * workgroup loops over an equal share of its class, so a grid of 8 per CU against 3 resident ran its last third at 2/3 of the
                     * machine (shading 83.2 -> 81.9 ms; 4 per CU: 97 ms); FTN_SHADE_GRID / FTN_SHADE_GRID_NOBSDF set it by hand */
                    const unsigned grid_k = knob("FTN_SHADE_GRID", 0), grid_k_nb = knob("FTN_SHADE_GRID_NOBSDF", 0);
#define FTN_SH2(T, M, E, mask) do { const unsigned per_cu = (M) == -2 ? (grid_k_nb ? grid_k_nb : 2u * shade_wgs_per_cu<T, M, E>()) : (grid_k ? grid_k : shade_wgs_per_cu<T, M, E>()); \
                                    hipLaunchKernelGGL((k_wf_shade<T, M, E>), dim3(std::min<unsigned>((unsigned)st->n_cu * per_cu, (W.n_paths + 255) / 256)), dim3(256), 0, stream, P, W, in_q, mask, first); } while (0)
#define FTN_SH(M, mask) do { if (tex) { if (env) FTN_SH2(true, M, true, mask); else FTN_SH2(true, M, false, mask); } \
                             else { if (env) FTN_SH2(false, M, true, mask); else FTN_SH2(false, M, false, mask); } } while (0)
                    FTN_SH(-2, 0x83u);
                    if (P.S.material_types & 1u) FTN_SH(0, 1u << 2);
                    if (P.S.material_types & 2u) FTN_SH(1, 1u << 3);
                    if (P.S.material_types & 4u) FTN_SH(2, 1u << 4);
                    if (P.S.material_types & 8u) FTN_SH(3, 1u << 5);
                    if (P.S.material_types & 16u) FTN_SH(4, 1u << 6);
#undef FTN_SH2
#undef FTN_SH
                }
            }
            span_end(e_shade, 2, stream);
            in_q ^= 1;
            if (sort_bits) {   /* order the two ray queues the next traces read (needs their lengths on the host: one small read-back) */
                WF_TRY(hipMemcpyAsync(st->host_counters, W.counters, 16 * 32 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                WF_TRY(hipStreamSynchronize(stream));
                polled = true;
#ifdef FTN_DRAIN_PROBE
                { const uint32_t* T = &st->host_counters[CTR(13) + 4 * (it & 7u)];
                  if (T[3]) fprintf(stderr, "[wf] any-hit launch of bounce %u: first wave dry at %.1f us, last wave dry at %.1f us, end %.1f us\n", it, (~T[1] - ~T[0]) * 0.01, (T[2] - ~T[0]) * 0.01, (T[3] - ~T[0]) * 0.01); }
#endif
                const size_t n = st->cap_paths;                        /* scratch: q_sorted is free until the next classify */
                uint32_t* const k_in = W.q_sorted + 4 * n, *const k_out = W.q_sorted + 6 * n;     /* [0,2n) sorted closest, [2n,4n) sorted any-hit, keys in / out */
                q_cl = W.q_closest; q_sh = W.q_shadow;
                const int e_sort = span_begin(stream);
                if ((rc = sort_ray_queue(st, P, W, false, W.q_closest, st->host_counters[CTR(2)], W.q_sorted, k_in, k_out, sort_bits, stream, &q_cl))) return rc;
                /* the any-hit queue stays in shading order: with the two-box kernel its sort (24 M entries per step) costs more than it returns */
                if (knob("FTN_WF_SORT_ANY", 0) && (rc = sort_ray_queue(st, P, W, true, W.q_shadow, st->host_counters[CTR(3)], W.q_sorted + 2 * n, k_in, k_out, sort_bits, stream, &q_sh))) return rc;
                span_end(e_sort, 3, stream);
            }
            if (it >= P.max_depth || polled) {   /* bounce max_depth has been shaded: poll whether anything (null-material pass-throughs) is left */
                if (!polled) {
                    WF_TRY(hipMemcpyAsync(st->host_counters, W.counters, 16 * 32 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                    WF_TRY(hipStreamSynchronize(stream));
                }
                if (it >= P.max_depth && count && knob("FTN_WF_DEBUG", 0)) fprintf(stderr, "[wf] wave steps so far: node %u leaf %u; lanes during node steps (x64): idle %u waiting-with-leaf %u\n", st->host_counters[CTR(6)], st->host_counters[CTR(7)], st->host_counters[CTR(8)], st->host_counters[CTR(9)]);
                n_active = st->host_counters[CTR(in_q == 0 ? 0 : 1)];
                if (n_active == 0) break;
            }
        }
        if (W.mis_any) mis_any_rays += st->host_counters[CTR(10)];   /* the last poll of the pass saw the pass's total */
        hipLaunchKernelGGL(k_wf_accumulate, dim3((n_slots + 255) / 256), dim3(256), 0, stream, P, W);
    }
    return FTN_OK;
    };
    rc = render_passes();
    { const int rc_flush = flush_events(); if (!rc) rc = rc_flush; }
    if (rc) return rc;
    WF_TRY(hipGetLastError());
    if (times) {
        times->trace_ms = trace_ms; times->trace_launches = trace_launches; times->mis_any_rays = mis_any_rays;
        times->any_ms = group_ms[1]; times->any_launches = group_n[1]; times->shade_ms = group_ms[2]; times->shade_launches = group_n[2]; times->sort_ms = group_ms[3];
    }
    return FTN_OK;
}

}  // namespace ftn
