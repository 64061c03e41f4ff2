/*
 * ftn_wf_common.h -- device-side declarations shared by the wavefront pipeline's translation units (ftn_wavefront.hip: generate /
 * classify / shade / accumulate and the reference-order traversal kernels; ftn_trace4.hip: the production traversal kernels over
 * 128-byte four-box records): path-state bits, the SoA buffers of a wavefront, queue helpers, the leaf primitive test.
 */
#ifndef FTN_WF_COMMON_H
#define FTN_WF_COMMON_H
#include "ftn_wavefront.h"

namespace ftn {

/* the bounce count takes 16 bits: PathIntegrator::max_depth is a u16 in the reference (path.rs:14), and ftn_render_device refuses more */
enum : uint32_t { PS_BOUNCE_MASK = 0xffffu, PS_SPECULAR = 1u << 16, PS_ALIVE = 1u << 17, PS_DIRECT = 1u << 18, PS_SHADOW = 1u << 19, PS_MIS = 1u << 20, PS_DELTA = 1u << 21, PS_MIS_ANY = 1u << 22 /* the MIS ray went through the any-hit kernel */ };
#define PS_DRAWS_SHIFT 23          /* rng_replay: draws made so far, 9 bits (wavefront_render enables it only when 5 + 8 * (max_depth + 1) fits) */
#define WF_MIS_BIT 0x80000000u

struct WfBuffers {
    uint32_t n_paths;           /* n_slots * samples; path id = slot * samples + sample */
    uint32_t n_slots;           /* tiles * 256 */
    uint32_t samples;           /* S */
    uint32_t first_sample;      /* 0-based index of this pass's first sample */
    /* rays / hits: index r in [0, n_paths) = continuation ray of path r, [n_paths, 2 n_paths) = MIS ray of path r - n_paths */
    float4 *ray;                /* 2 float4 per ray, adjacent (one 32-byte record = one line fetch when rays are read in sorted order): {o.xyz, -} {d.xyz, t_max} */
    float4* hit;                /* t, b0, b1, b2 */
    int* hit_prim;
    float4 *sh;                 /* shadow rays, per path, same 32-byte record */
    unsigned char* occluded;    /* per path */
    float4 *beta;               /* beta.xyz, bits(pstate) */
    float4 *rad;                /* L.xyz, bits(light index of the pending direct term) */
    float4* br;                 /* path integrator (NULL: not used): throughput and running radiance of a path side by side, br[2 p] = beta's record, br[2 p + 1] =
                                 * rad's -- the two are always read and written together, one scattered access instead of two; rad[] then only receives FINAL radiance */
    ulonglong2 *rng01, *rng23;  /* Xoshiro256+ state */
    float4 *pend0, *pend1, *pend2;   /* (Ld_light.xyz, weight) (f.xyz, pdf) (beta_prev.xyz, -) */
    float4* pd;                 /* path integrator (NULL: not used): the three pending terms of a path in one 64-byte record, pd[4 p + k] = pend<k>[p] (the fourth
                                 * float4 is padding: a record never straddles a 128-byte line) -- written together, read together */
    uint32_t pd_md, pd_occ;     /* 1 (with pd): the fourth float4 of a path's record is used too -- xyz = direction of its MIS ray (what the next event needs of that
                                 * ray when it went through the any-hit kernel), and the any-hit kernels store their results in the bytes of w (byte 0: the shadow
                                 * ray's, byte 1: the MIS ray's; store_occluded) instead of occluded[]: the next event finds them in the line it reads anyway */
    uint32_t *q_active[2], *q_closest, *q_shadow;   /* active-queue entries carry WF_Q_FIN / WF_Q_DEPTH in their top bits */
    uint32_t* q_sorted;         /* the active queue grouped by shading class (material-sorted shading) */
    uint32_t* cls;              /* per-class path counts, one 128-byte line each (CTR(k)) */
    uint32_t seg_cap;           /* capacity of one class segment of q_sorted */
    uint32_t* counters;         /* one 128-byte line each (CTR(i) = 32*i): 0 active A, 1 active B, 2 closest, 3 shadow, 4 head closest, 5 head shadow */
    uint32_t valid_per_sample;  /* camera samples per spp pass (sum of the tiles' pixel counts) */
    uint32_t* drain_sig; uint32_t drain_seq, drain_at;   /* closest-hit trace: where (signal memory) and what to store once a wave finds the queue dry (NULL: nobody waits) */
    /* rays the four-box kernels hand back to the reference-order kernels (a direction component of zero: see ftn_trace4.hip), queue
     * entries as in the queue they came from; counts at CTR(32) closest / CTR(33) any-hit, slice heads at CTR(40..47) / CTR(48..55) */
    uint32_t *q_exc_closest, *q_exc_any;
    /* DirectLightingIntegrator / WhittedIntegrator (k_wf_shade_dl): per-level terms of the nested product, level-major: dlA[d * n_paths + p] =
     * {local.rgb, pdf}, dlB = {f.rgb, |cos|}; whT[l * n_paths + p] = Whitted's term of light l (added if its shadow ray is unoccluded) */
    float4 *dlA, *dlB, *whT;
    /* textured scenes: the ray differentials that follow the specular chain (integrator/mod.rs:58-84), SoA: dfd[k * n_paths + p], k = 0..2 =
     * {rx_origin, rx_dir.x} {ry_origin, rx_dir.y} {ry_dir, rx_dir.z}; level 0 rebuilds the camera's from the sample key instead */
    float4* dfd;
    /* FTN_SAMPLER_TILE_SERIAL on the queues (the reference's RandomSampler: ONE Xoshiro stream per tile, random.rs:61-67): path id = tile,
     * one camera sample per tile in flight; a path that has nothing pending any more is marked by the shade kernels and k_wf_serial_advance
     * adds it to the film and starts the tile's next sample from the stream position the path left */
    uint32_t serial;
    uint2* ser_cursor;          /* per tile: {pixel index inside the tile (row-major over its extent), sample index} of the sample in flight */
    float2* ser_pfilm;          /* ... and its film position */
    unsigned char* ser_retired;
    uint32_t rng_replay;        /* 1: a path's Xoshiro stream is not carried in rng01 / rng23 but re-created from its sample key and the number of draws made so far
                                 * (bits 23-31 of the path-state word: PS_DRAWS_SHIFT): 64 bytes less per shading event, ~16 instructions per replayed draw */
    uint32_t gen_blocks;        /* 1: k_wf_generate queues the pixels of a full 16 x 16 tile in 2 x 2 blocks (Morton order) instead of rows: the 64 rays of a
                                 * wave of the first trace are then a 2 x 2 pixel block at 16 spp, not a 4 x 1 strip.  Queue ORDER only: no result depends on it */
    uint32_t mis_any;           /* 1: MIS rays toward an infinite light only need hit / miss -> any-hit kernel (off in the counting build, whose node tallies must equal the reference's closest-hit walk) */
};

/* flags in an ACTIVE-queue entry (path ids stay below 2^28: wavefront_render): what k_wf_classify needs to know about the path without
 * touching its state */
#define WF_Q_FIN 0x80000000u        /* the path has ended and only comes back for its pending direct term (class 0) */
#define WF_Q_DEPTH 0x40000000u      /* alive, but its bounce count has reached max_depth: emission only, whatever it hits (class 1) */
#define WF_Q_ID_MASK 0x0fffffffu

#define CTR(i) ((i) * 32)
__device__ inline uint32_t lane_id() { return threadIdx.x & 63u; }
/* wave-level queue append: one atomic per wave (ballot / popc compaction) */
__device__ inline void wave_push(bool pred, uint32_t value, uint32_t* queue, uint32_t* counter) {
    const unsigned long long m = __ballot(pred);
    if (m == 0) return;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    if (pred) queue[base + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull))] = value;
}

/* block-level queue append: ballot per wave, ONE global atomic per workgroup and queue (same-address atomics retire at
 * ~88 per microsecond on this part, so per-wave appends to one word were the bottleneck of generate/shade) */
template <int NQ>
__device__ inline void block_push(const bool (&pred)[NQ], const uint32_t (&value)[NQ], uint32_t* const (&queue)[NQ], uint32_t* const (&counter)[NQ]) {
    __shared__ uint32_t s_cnt[NQ][4];
    __shared__ uint32_t s_base[NQ];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    unsigned long long m[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) { m[q] = __ballot(pred[q]); if (lane == 0) s_cnt[q][wave] = (uint32_t)__popcll(m[q]); }
    __syncthreads();
    if (threadIdx.x < NQ) {
        const uint32_t q = threadIdx.x;
        const uint32_t tot = s_cnt[q][0] + s_cnt[q][1] + s_cnt[q][2] + s_cnt[q][3];
        s_base[q] = tot ? atomicAdd(counter[q], tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        if (pred[q]) {
            uint32_t off = s_base[q];
            for (uint32_t w = 0; w < wave; w++) off += s_cnt[q][w];
            if (queue[q]) queue[q][off + (uint32_t)__popcll(m[q] & ((1ull << lane) - 1ull))] = value[q];
        }
    }
    __syncthreads();   /* s_cnt / s_base are reused by the next round */
}

/* result of an any-hit ray; slot as load_queued_ray<true> gives it ([0, n_paths) shadow rays, [n_paths, 2 n_paths) MIS rays) */
__device__ inline void store_occluded(const WfBuffers& W, uint32_t slot, bool occ) {
    if (W.pd_occ) {
        if (!occ) return;        /* the shading event that queued the ray wrote the record with w = 0: "not occluded" is already there */
        const bool mis = slot >= W.n_paths;
        const uint32_t p = mis ? slot - W.n_paths : slot;
        ((unsigned char*)(W.pd + 4 * (size_t)p + 3))[12 + (mis ? 1 : 0)] = occ ? 1 : 0;
    } else W.occluded[slot] = occ ? 1 : 0;
}

/* A queue entry is a path id; WF_MIS_BIT marks the path's MIS ray (record n_paths + id) instead of its continuation ray (closest-hit
 * queue) or its shadow ray (any-hit queue).  *slot: index of the ray's result (hit[] / hit_prim[] for closest hits, occluded[] for
 * any-hit: shadow results at [0, n_paths), MIS-ray results at [n_paths, 2 n_paths)). */
template <bool ANY>
__device__ inline void load_queued_ray(const WfBuffers& W, uint32_t rid, float4* a, float4* b, uint32_t* slot) {
    const uint32_t pid = rid & ~WF_MIS_BIT;
    if (ANY && !(rid & WF_MIS_BIT)) { *a = W.sh[2 * (size_t)pid]; *b = W.sh[2 * (size_t)pid + 1]; *slot = pid; }
    else { const uint32_t r = (rid & WF_MIS_BIT) ? pid + W.n_paths : pid; *a = W.ray[2 * (size_t)r]; *b = W.ray[2 * (size_t)r + 1]; *slot = r; }
}

/* one primitive of a leaf against the ray: Triangle::intersect's hit test (triangle.rs:183-268) with the per-ray permutation
 * (kz) and shear (sx, sy, sz) hoisted, or Sphere::intersect */
template <bool SPHERES>
__device__ inline bool prim_hit(const DScene& S, uint32_t prim, float4 g0, float4 g1, float4 g2, V3 o, V3 dorig, float t_max, int kz, float sx, float sy, float sz,
                                float* t_out, float* b0o, float* b1o, float* b2o) {
    const uint32_t fl = __float_as_uint(g0.w);
    float t = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f; bool hh = false;
    if (SPHERES && (fl & GF_KIND_SPHERE)) {
        DRay r; r.o = o; r.d = dorig; r.t_max = t_max; r.time = 0.0f;
        hh = sphere_intersect<false>(S.spheres[__float_as_uint(g1.w)], r, &t, nullptr);
    } else {
        /* Triangle::intersect hit test (triangle.rs:183-268) with the per-ray constants hoisted */
        /* permute_point(p - o, kx, ky, kz) with (kx, ky, kz) = (kz+1, kz+2, kz) mod 3, as selects on registers */
        const bool k0 = kz == 0, k1 = kz == 1;
#define FTN_PERM(v) V3(k0 ? (v).y : (k1 ? (v).z : (v).x), k0 ? (v).z : (k1 ? (v).x : (v).y), k0 ? (v).x : (k1 ? (v).y : (v).z))
        const V3 op = FTN_PERM(o);
        V3 p0t = FTN_PERM(g0), p1t = FTN_PERM(g1), p2t = FTN_PERM(g2);
#undef FTN_PERM
        p0t = V3(p0t.x - op.x, p0t.y - op.y, p0t.z - op.z);
        p1t = V3(p1t.x - op.x, p1t.y - op.y, p1t.z - op.z);
        p2t = V3(p2t.x - op.x, p2t.y - op.y, p2t.z - op.z);
        p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
        p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
        p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
        float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
        float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
        float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
        if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
            e0 = (float)((double)p1t.x * (double)p2t.y - (double)p1t.y * (double)p2t.x);
            e1 = (float)((double)p2t.x * (double)p0t.y - (double)p2t.y * (double)p0t.x);
            e2 = (float)((double)p0t.x * (double)p1t.y - (double)p0t.y * (double)p1t.x);
        }
        const float det = e0 + e1 + e2;
        if (!(sign_pos(e0) != sign_pos(e1) || sign_pos(e1) != sign_pos(e2)) && det != 0.0f) {
            p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
            const float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
            if (!((det < 0.0f && (t_scaled >= 0.0f || t_scaled < t_max * det)) || (det > 0.0f && (t_scaled <= 0.0f || t_scaled > t_max * det)))) {
                const float inv_det = 1.0f / det;
                b0 = e0 * inv_det; b1 = e1 * inv_det; b2 = e2 * inv_det; t = t_scaled * inv_det;
                const float max_zt = fmax_(fmax_(fabsf(p0t.z), fabsf(p1t.z)), fabsf(p2t.z));
                const float delta_z = gamma_n(3) * max_zt;
                const float max_xt = fmax_(fmax_(fabsf(p0t.x), fabsf(p1t.x)), fabsf(p2t.x));
                const float max_yt = fmax_(fmax_(fabsf(p0t.y), fabsf(p1t.y)), fabsf(p2t.y));
                const float delta_x = gamma_n(5) * (max_xt + max_zt), delta_y = gamma_n(5) * (max_yt + max_zt);
                const float delta_e = 2.0f * (gamma_n(2) * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
                const float max_e = fmax_(fmax_(fabsf(e0), fabsf(e1)), fabsf(e2));
                const float delta_t = 3.0f * (gamma_n(3) * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * fabsf(inv_det);
                hh = !(t <= delta_t);
                if (hh && (fl & GF_HAS_UVS) && tri_uv_degenerate_reject(S, (int)prim, V3(g0.x, g0.y, g0.z), V3(g1.x, g1.y, g1.z), V3(g2.x, g2.y, g2.z))) hh = false;
            }
        }
    }
    *t_out = t; *b0o = b0; *b1o = b1; *b2o = b2;
    return hh;
}


/* ------------------------------------------------------------------ production traversal kernels over four-box records (ftn_trace4.hip) */
struct Trace4Plan {
    uint32_t entries_closest, entries_any;      /* stack levels kept in LDS per lane */
    uint32_t spill_closest, spill_any;          /* further levels in the per-lane global spill area (0: everything fits in LDS) */
    size_t lds_closest, lds_any;                /* bytes of dynamic LDS per 256-thread workgroup */
    unsigned grid_closest, grid_any;            /* persistent workgroups: CUs x workgroups per CU */
    bool oct_ok; uint32_t entries_oct, spill_oct; size_t lds_oct; unsigned grid_oct;            /* k_wf_trace8_any (eight-box occlusion records) */
};
enum : int { T4K_CLOSEST = 0, T4K_ANY = 1, T4K_ANY_OCT = 2 };
Trace4Plan trace4_plan(const DScene& S, int n_cu, uint32_t knob_entries_closest, uint32_t knob_entries_any, uint32_t knob_wg_closest, uint32_t knob_wg_any, uint32_t knob_wg_oct, uint32_t knob_entries_oct);
void launch_trace4(int kind, bool count, bool spheres, unsigned grid, size_t lds, uint32_t lds_entries, void* spill, hipStream_t stream, const DScene& S, const WfBuffers& W,
                   const uint32_t* queue, const uint32_t* count_ptr, uint32_t* head, DevStats* stats, uint32_t refill, uint32_t leaf_batch, uint32_t chunk, uint32_t burst, uint32_t any_policy, uint32_t spill_levels);

}  // namespace ftn
#endif
