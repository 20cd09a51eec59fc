/*
 * rt_kernels.h -- the HIP kernels behind the C-ABI of include/rtamd.h (gfx950, wave64), in rt_device.h terms.
 * 256-thread workgroups except k_render, which runs one wave per workgroup.
 *   k_render         one WAVE per pixel, one lane per sample of a 64-sample chunk; each lane traces
 *                    its whole path tree; the wave then adds its lanes' colours in sample order
 *                    (the reference's float summation order, R/main.cpp:185-193) and lane 0
 *                    stores the pixel.  Lanes of a wave share a pixel, so they follow almost the
 *                    same path and the scalar-cached scene tables stay wave-uniform.  With 32 or
 *                    fewer samples per pixel a wave takes 64 / spp pixels instead.
 *   k_tile_class, k_tile_place   before each render: order the tiles by their deepest path tree.
 *   k_intersect      one lane per ray: Scene::computeIntersection.
 *   k_mesh_intersect one lane per ray: Mesh::intersect of one mesh.
 *   k_trace          one lane per ray: Scene::rayTrace under the counter RNG.
 *   k_generate       one lane per (u,v): screen_space_to_world_space_ray.
 * Kept apart from the host code (rt_api.hip) so that one kernel can be compiled on its own (tools/one_kernel.sh).
 */
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>

#include "rt_device.h"
/* ======================================================================================== */
/*                                         kernels                                          */
/* ======================================================================================== */

struct RenderParams {
    uint32_t w, h, spp, seed;
    uint32_t tile_w, tile_h, tiles_x, n_tiles;
    uint32_t rank, world, n_local_tiles, sample0; /* first sample of this pass */
    uint32_t spp_total, pad[3];                   /* samples per pixel of the finished image */
};

/* Work list of the pixels whose path trees are deep: k_tile_class fills it, k_render skips the listed pixels,
 * k_render_wide traces them with several lanes per path and k_resolve_deep adds their samples up.  All pointers null = off. */
struct DeepList {
    uint32_t* head;     /* [0] pixels appended (may run past cap: readers clamp), [1] next item of k_render_wide's queue */
    uint32_t* pixels;   /* cap entries: index of the pixel in the rank's tile buffer = local_tile * pixels per tile + position in tile */
    uint8_t* flags;     /* per pixel of the tile buffer: 1 = listed */
    float* colors;      /* cap x spp of the pass x 3: the colour of every sample of the listed pixels */
    uint32_t* tree;     /* n_workers x RT_TREE_NODES x RT_TREE_WORDS: the path trees a worker wave is expanding */
    uint32_t cap, min_nodes; /* a pixel is listed when the probe of its centre ray counts at least min_nodes path nodes */
    uint32_t n_workers, pad;
};

RT_DEV Cnt new_cnt(uint32_t* stack, uint32_t stack_stride, float* frames = nullptr, uint32_t frame_stride = 0) {
    Cnt c;
    c.rays = 0; c.nodes = 0; c.bvh_nodes = 0; c.bvh_tris = 0; c.skipped = 0;
    c.stack = stack; c.stack_stride = stack_stride; c.frames = frames; c.frame_stride = frame_stride;
    return c;
}

/* sum of a per-lane counter over the wave, in every lane */
RT_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

/* The same sum, in a wave-uniform register, by data-parallel-primitive adds (an inclusive scan within each row of 16 lanes,
 * then the row totals carried into the next rows: lane 63 ends up with the total): six VALU instructions where the
 * shuffles above are six round trips through the LDS crossbar. */
RT_DEV uint32_t wave_total(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true); /* row_shr:1, lanes without a source add 0 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true); /* row_shr:2 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true); /* row_shr:4 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true); /* row_shr:8 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); /* row_bcast:15 into rows 1 and 3 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); /* row_bcast:31 into rows 2 and 3 */
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

/* one set of atomics per workgroup, into the workgroup's slot (see RT_COUNTER_SLOTS): a single
 * shared address would serialise ~6 M atomics per frame at ~12 ns each.  ONE_WAVE: the workgroup is a single wave whose
 * lanes are all here (k_render): no LDS partial sums and no barriers, lane k adds field k. */
template <int M, bool ONE_WAVE = false>
RT_DEV void flush_counters(uint64_t* counters, const Cnt& c, uint32_t samples) {
    if (!counters) return;
    constexpr int N = RT_COUNTER_FIELDS;
    if (ONE_WAVE) {
        uint32_t t[N] = {wave_total(c.rays), wave_total(samples), wave_total(c.nodes), 0u, 0u, wave_total(c.skipped)};
#ifdef RT_EXP_COUNT_OPEN
        if (true) { t[3] = wave_total(c.bvh_nodes); t[4] = wave_total(c.bvh_tris); }
#else
        if (M & RT_M_BVH) { t[3] = wave_total(c.bvh_nodes); t[4] = wave_total(c.bvh_tris); }
#endif
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) mine = (int)threadIdx.x == k ? t[k] : mine;
        if (threadIdx.x < N && mine) {
            uint64_t* slot = counters + (size_t)(blockIdx.x % RT_COUNTER_SLOTS) * RT_COUNTER_STRIDE;
            atomicAdd((unsigned long long*)&slot[threadIdx.x], (unsigned long long)mine);
        }
        return;
    }
    __shared__ uint32_t part[N];
    if (threadIdx.x < N) part[threadIdx.x] = 0;
    __syncthreads();
    uint32_t v[N] = {wave_sum(c.rays), wave_sum(samples), wave_sum(c.nodes), 0u, 0u, wave_sum(c.skipped)};
    if (M & RT_M_BVH) { v[3] = wave_sum(c.bvh_nodes); v[4] = wave_sum(c.bvh_tris); } /* only these variants traverse one */
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < N; ++k)
            if (v[k]) atomicAdd(&part[k], v[k]);
    __syncthreads();
    if (threadIdx.x < N && part[threadIdx.x]) {
        uint64_t* slot = counters + (size_t)(blockIdx.x % RT_COUNTER_SLOTS) * RT_COUNTER_STRIDE;
        atomicAdd((unsigned long long*)&slot[threadIdx.x], (unsigned long long)part[threadIdx.x]);
    }
}

#ifndef RT_RENDER_WAVES
/* Waves (pixels) per workgroup of the render kernel.  One: a workgroup holds its wave slots until its slowest
 * wave is done, and next to a glass-rim pixel that takes 3 ms the three neighbours' slots would sit idle
 * (measured: 27.7 -> 27.2 ms on the full frame, 3.74 -> 3.61 ms on the slowest of eight ranks). */
#define RT_RENDER_WAVES 1
#if RT_RENDER_WAVES != 1
#error "the worker workgroups of k_render are single waves"
#endif
#endif
#ifndef RT_FRAME_LDS
#define RT_FRAME_LDS 5 /* levels of the path-frame stack kept in LDS by the render variants without a BVH: 5 x 7 x 256 B = 8960 B per wave */
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4 /* waves per SIMD the render variants are compiled for (<= 128 VGPRs) */
#endif
/* The variant of the scenes whose objects all sit in one leaf and that have no BVH (mode 0: every built-in scene) waits
 * for its own instructions more than it computes, so it runs best with more, smaller waves: shadow rays in batches of
 * two instead of four (RT_SB0) fit 80 VGPRs, i.e. six waves per SIMD, with three LDS levels of path frames
 * (3 x 7 x 256 B x 24 waves = 126 KB of the CU's 160 KB).  Measured on the Cornell 1080p frame, batch / waves:
 * 4 / 4 21.6 ms, 2 / 4 22.6, 4 / 5 22.7, 2 / 5 21.2, 2 / 6 20.45, 2 / 7 20.6, 2 / 8 21.3, 1 / 6 21.8, 1 / 8 22.6. */
#ifndef RT_MIN_WAVES0
#define RT_MIN_WAVES0 6
#endif
/* Scenes with a triangle BVH and a root leaf (mode 2) also take batches of two, at their four waves: a third of the register
 * spills (29 against 68 VGPRs), 1 M-triangle frame 587 -> 581 ms, 10 M-triangle frame in four passes 74.7 -> 71.4 ms. */
#ifndef RT_SB2
#define RT_SB2 2
#endif
#ifndef RT_FRAME_LDS0
#define RT_FRAME_LDS0 3
#endif
#ifdef RT_WAVES_PER_EU
#define RT_OCC __attribute__((amdgpu_waves_per_eu(RT_WAVES_PER_EU, RT_WAVES_PER_EU)))
#else
#define RT_OCC
#endif
/* ---- launch order of the tiles: expensive tiles first -------------------------------------
 * A pixel that looks into the glass sphere traces a tree of up to 2^11 rays per sample: its wave
 * runs for ~1 ms, near the sphere's rim for more than 3 ms, where an ordinary one runs for ~45 us.
 * When such a wave is handed out late it finishes alone after the rest of the grid (measured: one
 * rim pixel started at 2 ms stretched a 4.2 ms launch of one eighth of the frame to 5.3 ms).  So
 * before each render one wave per tile walks the path tree of the centre ray of every pixel of
 * the tile without shading it, and the tiles are launched in the order of their deepest tree
 * (capped at RT_PROBE_CAP nodes, where an ordinary glass pixel has 16-21; counting sort in
 * k_tile_place).  The order only changes WHEN a tile is rendered: pixel values, RNG streams and
 * the output layout depend on the tile id alone.
 * Workspace (uint32): [0, n) key of local tile i (0 = deepest), [n, 2n) the order. */
#define RT_ORDER_KEYS (RT_PROBE_CAP + 1)
template <int M, int T = 256> /* T: threads of the workgroup, 64 for tiles of at most 64 pixels; the traversal stack in LDS is sized by it */
__global__ __launch_bounds__(T) void k_tile_class(DeviceScene sc, const DCamera* __restrict__ cam, RenderParams rp, uint32_t* __restrict__ ws, DeepList dl) {
    /* The probe only has to say where path trees are deep, which the spheres decide.  In a scene whose root is a leaf
     * it therefore leaves the meshes out: no BVH is traversed, and a 10 M-triangle scene is probed as fast as the bare box. */
    constexpr int P = (M == RT_M_BVH) ? 0 : M;
    if (M == RT_M_BVH) sc.n_meshes = 0;
    __shared__ uint32_t s_stack[RT_STACK_LDS * T];
    __shared__ uint32_t s_deepest, s_bounces;
    const uint32_t local_tile = blockIdx.x; /* one workgroup per tile, one thread per pixel (a 16 x 16 tile) */
    const uint32_t tile = rp.rank + local_tile * rp.world;
    const uint32_t x0 = rt_tile_column(tile, rp.tiles_x) * rp.tile_w, y0 = (tile / rp.tiles_x) * rp.tile_h;
    const uint32_t px_per_tile = rp.tile_w * rp.tile_h;
    if (threadIdx.x == 0) { s_deepest = 0; s_bounces = 0; }
    __syncthreads();
    auto probe = [&](uint32_t x, uint32_t y, float fx, float fy) -> uint32_t {
        float u = ((float)(int)x + fx) / (float)(int)rp.w;
        float v = ((float)(int)y + fy) / (float)(int)rp.h;
        f3 o, d;
        primary_ray(*cam, u, v, o, d);
        Cnt cnt = new_cnt(s_stack + threadIdx.x, T);
        Rng rng; rng.key = 0; rng.at(0);
        (void)trace_path<P, true>(sc, o, norm3(d), rng, cnt);
        return cnt.nodes;
    };
    uint32_t deepest = 0;
    for (uint32_t i = threadIdx.x; i < px_per_tile; i += T) { /* the centre of every pixel */
        const uint32_t x = x0 + i % rp.tile_w, y = y0 + i / rp.tile_w;
        uint32_t nodes = 0;
        if (x < rp.w && y < rp.h) nodes = probe(x, y, 0.5f, 0.5f);
        deepest = max(deepest, nodes);
        if (dl.flags) dl.flags[(uint64_t)local_tile * px_per_tile + i] = (uint8_t)nodes; /* for the second round */
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) deepest = max(deepest, (uint32_t)__shfl_xor((int)deepest, off, 64));
    if ((threadIdx.x & 63) == 0) { atomicMax(&s_deepest, deepest); if (deepest >= 2u) s_bounces = 1u; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t key = (RT_ORDER_KEYS - 1) - min(s_deepest, (uint32_t)(RT_ORDER_KEYS - 1)); /* key 0 = deepest */
        ws[local_tile] = key;
    }
    if (!dl.flags) return;
    /* Deep list.  A tile some of whose pixels look at a mirror or glass sphere also probes the corners of its pixels: the
     * samples of a pixel on the rim of a sphere differ, and one deep sample makes the pixel's wave a long one. */
    const bool corners = s_bounces != 0u; /* the same for the whole workgroup */
    for (uint32_t i0 = 0; i0 < px_per_tile; i0 += T) { /* every thread of the workgroup makes every trip */
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t x = x0 + i % rp.tile_w, y = y0 + i / rp.tile_w;
        const bool valid = i < px_per_tile && x < rp.w && y < rp.h;
        uint32_t nodes = 0;
        if (valid) {
            nodes = dl.flags[(uint64_t)local_tile * px_per_tile + i];
            if (corners && nodes < dl.min_nodes) nodes = max(nodes, probe(x, y, 0.05f, 0.05f));
            if (corners && nodes < dl.min_nodes) nodes = max(nodes, probe(x, y, 0.95f, 0.05f));
            if (corners && nodes < dl.min_nodes) nodes = max(nodes, probe(x, y, 0.05f, 0.95f));
            if (corners && nodes < dl.min_nodes) nodes = max(nodes, probe(x, y, 0.95f, 0.95f));
        }
        /* one atomic per wave reserves the list entries of its deep pixels */
        const bool deep = valid && nodes >= dl.min_nodes;
        const uint64_t mask = __ballot(deep);
        uint32_t first = 0;
        if (mask) {
            const uint32_t lane = threadIdx.x & 63;
            if (lane == (uint32_t)__builtin_ctzll(mask)) first = atomicAdd(&dl.head[0], (uint32_t)__popcll(mask));
            first = (uint32_t)__shfl((int)first, (int)__builtin_ctzll(mask), 64);
        }
        const uint32_t entry = first + wave_rank(mask, threadIdx.x & 63);
        const bool listed = deep && entry < dl.cap;
        if (listed) dl.pixels[entry] = local_tile * px_per_tile + i;
        if (i < px_per_tile) dl.flags[(uint64_t)local_tile * px_per_tile + i] = listed ? 1 : 0;
    }
}

/* One workgroup: histogram of the keys, exclusive scan, placement, all with LDS atomics (a
 * grid-wide version spends 80 us on 8100 global atomics to two or three addresses). */
__global__ __launch_bounds__(1024) void k_tile_place(uint32_t* __restrict__ ws, uint32_t n) {
    __shared__ uint32_t s_count[RT_ORDER_KEYS], s_cursor[RT_ORDER_KEYS];
    if (threadIdx.x < RT_ORDER_KEYS) s_count[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024) atomicAdd(&s_count[ws[i]], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int k = 0; k < RT_ORDER_KEYS; ++k) { s_cursor[k] = run; run += s_count[k]; }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024) ws[n + atomicAdd(&s_cursor[ws[i]], 1u)] = i;
}

/* ---- path trees expanded level by level: the worker waves of k_render ------------------------------------------
 * A pixel on the deep list has path trees of hundreds of nodes per sample (glass seen through glass).  Walked depth
 * first by one lane per sample they are chains of hundreds of dependent closest-hit and shadow queries, and the wave
 * that holds them is still running when the rest of the launch is done.  A worker wave instead expands the trees of up
 * to RT_TREE_BATCH samples of such pixels breadth first -- the recursion of R/src/Scene.h:596-713 has depth <= 11, so
 * 11 rounds of closest-hit queries, one lane per node, find every node -- and then shades all nodes that need it side
 * by side.  What the recursion's order decides is kept:
 *   random numbers  a node that was hit draws 32 numbers per quad light (R/src/Scene.h:369-370), nodes in depth-first
 *                   order, reflected subtree before refracted (R/src/Scene.h:694,701): node n's first draw is
 *                   2 + Q * (number of hit nodes before n in that order), from subtree sizes added up level by level;
 *   colours         combined upwards with the expressions of R/src/Scene.h:655,705, children first.
 * Every value is computed by the routines trace_path uses (closest_rec, finish_hit, bounce_rays, shade), so a pixel
 * gets the same floats whichever way it is rendered.  Children are appended to the next level with one ballot per
 * chunk of 64 nodes.  Node fields live in this worker's slice of dl.tree (field f of node i at
 * [f * RT_TREE_NODES + i]); a batch whose trees do not fit is cut in halves (one sample's tree has at most 2047 nodes).
 * Workers take items (RT_TREE_BATCH consecutive samples of the pass's deep samples, listed pixel by listed pixel: the
 * samples of one item belong to one pixel or to pixels the probe listed next to each other) from a counter until none is
 * left; they are the first workgroups of the grid, so the long items start first. */
#define RT_TREE_NODES 2048
#define RT_TREE_WORDS 14
#ifndef RT_TREE_BATCH
#define RT_TREE_BATCH 32
#endif
#ifndef RT_TREE_WORKERS
#define RT_TREE_WORKERS 4096
#endif
enum { TF_A0 = 0, TF_META = 6, TF_OBJ = 7, TF_F = 8, TF_C1 = 9, TF_C2 = 10, TF_SIZE = 11, TF_PRE = 12, TF_LIST = 13 };
/* fields 0..5: the node's ray (origin, direction) until it is traced, then hit position and normal, then (0..2) its colour.
 * meta = kind | hit type << 4 | sample of the batch << 8; kind 0 miss, 1 hit without children, 2 mirror, 3 glass with
 * both children, 4 glass with the reflected child only */
RT_DEV void tree_sync() { /* this wave's stores to the tree are visible to its later loads, whichever lane made them */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <int M>
RT_DEV bool tree_batch(const DeviceScene& sc, const DCamera* __restrict__ cam, const RenderParams& rp, uint32_t* __restrict__ mem, uint32_t* s_lv,
                       const DeepList& dl, uint32_t f_first, uint32_t n_samples, uint32_t draws_per_node, Cnt& cnt) {
    const uint32_t lane = threadIdx.x & 63;
    /* sample f of the pass's deep samples = sample f % spp of listed pixel f / spp; pixel index and sample number of it */
    auto sample_of = [&](uint32_t f, uint32_t& pixel, uint32_t& s) {
        const uint32_t pix = dl.pixels[f / rp.spp], px_per_tile = rp.tile_w * rp.tile_h;
        const uint32_t tile = rp.rank + (pix / px_per_tile) * rp.world, in_tile = pix % px_per_tile;
        const uint32_t x = rt_tile_column(tile, rp.tiles_x) * rp.tile_w + in_tile % rp.tile_w;
        const uint32_t y = (tile / rp.tiles_x) * rp.tile_h + in_tile / rp.tile_w;
        pixel = x + y * rp.w;
        s = rp.sample0 + f % rp.spp;
    };
    auto fld = [&](int f, uint32_t i) -> uint32_t& { return mem[(uint32_t)f * RT_TREE_NODES + i]; };
    auto put3 = [&](int f, uint32_t i, f3 v) { fld(f, i) = __float_as_uint(v.x); fld(f + 1, i) = __float_as_uint(v.y); fld(f + 2, i) = __float_as_uint(v.z); };
    auto get3 = [&](int f, uint32_t i) { return mk3(__uint_as_float(fld(f, i)), __uint_as_float(fld(f + 1, i)), __uint_as_float(fld(f + 2, i))); };
    /* roots: R/main.cpp:186-190 */
    if (lane < n_samples) {
        uint32_t pixel, smp;
        sample_of(f_first + lane, pixel, smp);
        Rng rng; rng.key = rt_rng_key(rp.seed, pixel, smp); rng.at(0);
        float u = ((float)(int)(pixel % rp.w) + rng.unit()) / (float)(int)rp.w;
        float v = ((float)(int)(pixel / rp.w) + rng.unit()) / (float)(int)rp.h;
        f3 o, d;
        primary_ray(*cam, u, v, o, d);
        put3(TF_A0, lane, o);
        put3(TF_A0 + 3, lane, norm3_unit(d));
        fld(TF_META, lane) = lane << 8;
    }
    tree_sync();
    /* 1. expansion, one level per round */
    uint32_t ls = 0, le = n_samples, depth = 0;
    if (lane == 0) s_lv[0] = 0;
    for (; ls < le; ++depth) {
        uint32_t next = le;
        for (uint32_t c = ls; c < le; c += 64) {
            const uint32_t i = c + lane;
            const bool active = i < le;
            uint32_t nchild = 0, kind = 0, meta = 0, c1 = 0, c2 = 0;
            Bounce bn;
            bn.kind = 0; bn.F = 0.f;
            bn.ro = bn.rd = bn.to = bn.td = mk3(0.f, 0.f, 0.f);
            if (active) {
                meta = fld(TF_META, i);
                const f3 o = get3(TF_A0, i), d = get3(TF_A0 + 3, i);
                HitRec rec = closest_rec<M>(sc, o, d, cnt);
                if (rec.type >= 0) {
                    FullHit h = finish_hit(sc, rec, o, d);
                    put3(TF_A0, i, h.p);
                    put3(TF_A0 + 3, i, h.n);
                    fld(TF_OBJ, i) = (uint32_t)h.obj;
                    meta |= (uint32_t)h.type << 4;
                    kind = 1;
                    if (depth < RT_MAX_BOUNCES && h.type == RT_SPHERE) {
                        const int mat = ldc(&sc.spheres[h.obj].mat), mtype = ldc(&sc.materials[mat].type);
                        if (mtype != RT_MAT_DIFFUSE) {
                            bn = bounce_rays(sc, mtype, mat, h.p, h.n, d);
                            kind = 2 + bn.kind;
                            nchild = bn.kind == 1u ? 2u : 1u;
                            if (mtype == RT_MAT_GLASS) { /* its Phong colour is never used: only what Scene::phong would have counted */
                                Rng none; none.key = 0; none.at(0);
                                skip_shade(sc, none, cnt);
                            }
                        }
                    }
                }
            }
            const uint64_t m1 = __ballot(nchild >= 1u), m2 = __ballot(nchild == 2u);
            const uint32_t n1 = (uint32_t)__popcll(m1), total = n1 + (uint32_t)__popcll(m2);
            if (next + total > RT_TREE_NODES) return false; /* wave-uniform: the batch does not fit */
            if (nchild >= 1u) {
                c1 = next + wave_rank(m1, lane);
                put3(TF_A0, c1, bn.ro);
                put3(TF_A0 + 3, c1, bn.rd);
                fld(TF_META, c1) = meta & 0xffffff00u;
            }
            if (nchild == 2u) {
                c2 = next + n1 + wave_rank(m2, lane);
                put3(TF_A0, c2, bn.to);
                put3(TF_A0 + 3, c2, bn.td);
                fld(TF_META, c2) = meta & 0xffffff00u;
            }
            if (active) {
                fld(TF_META, i) = meta | kind;
                fld(TF_F, i) = __float_as_uint(bn.F);
                fld(TF_C1, i) = c1;
                fld(TF_C2, i) = c2;
            }
            next += total;
        }
        ls = le; le = next;
        if (lane == 0) s_lv[depth + 1] = ls;
        tree_sync();
    }
    const uint32_t n_nodes = le, n_levels = depth; /* level l = nodes [s_lv[l], s_lv[l + 1]), s_lv[n_levels] = n_nodes */
    wave_lds_sync();
#ifdef RT_WAVE_TIMES
    const unsigned long long t_expanded = wall_clock64();
#endif
    /* 2. hit nodes per subtree, deepest level first */
    for (uint32_t l = n_levels; l-- > 0;) {
        const uint32_t b = s_lv[l], e = s_lv[l + 1];
        for (uint32_t i = b + lane; i < e; i += 64) {
            const uint32_t kind = fld(TF_META, i) & 15u, c1 = fld(TF_C1, i), c2 = fld(TF_C2, i);
            fld(TF_SIZE, i) = (kind != 0u ? 1u : 0u) + (c1 ? fld(TF_SIZE, c1) : 0u) + (c2 ? fld(TF_SIZE, c2) : 0u);
        }
        tree_sync();
    }
    /* 3. position among the hit nodes of the sample in depth-first order, top level first */
    if (lane < n_samples) fld(TF_PRE, lane) = 0;
    tree_sync();
    for (uint32_t l = 0; l < n_levels; ++l) {
        const uint32_t b = s_lv[l], e = s_lv[l + 1];
        for (uint32_t i = b + lane; i < e; i += 64) {
            const uint32_t c1 = fld(TF_C1, i), c2 = fld(TF_C2, i), pre = fld(TF_PRE, i);
            if (c1) fld(TF_PRE, c1) = pre + 1u;
            if (c2) fld(TF_PRE, c2) = pre + 1u + fld(TF_SIZE, c1);
        }
        tree_sync();
    }
    /* 4. the nodes Scene::phong's result is used of (every hit that is not a glass node with children), compacted */
    uint32_t n_list = 0;
    for (uint32_t c = 0; c < n_nodes; c += 64) {
        const uint32_t i = c + lane;
        const uint32_t kind = i < n_nodes ? (fld(TF_META, i) & 15u) : 0u;
        const bool want = kind == 1u || kind == 2u;
        const uint64_t m = __ballot(want);
        if (want) fld(TF_LIST, n_list + wave_rank(m, lane)) = i;
        n_list += (uint32_t)__popcll(m);
    }
    tree_sync();
    /* 5. shade them, one lane per node */
    for (uint32_t c = 0; c < n_list; c += 64) {
        if (c + lane < n_list) {
            const uint32_t i = fld(TF_LIST, c + lane);
            const uint32_t meta = fld(TF_META, i);
            FullHit h;
            h.hit = true; h.type = (int)((meta >> 4) & 15u); h.obj = (int)fld(TF_OBJ, i); h.tri = -1; h.t = 0.f;
            h.p = get3(TF_A0, i); h.n = get3(TF_A0 + 3, i);
            uint32_t pixel, smp;
            sample_of(f_first + (meta >> 8), pixel, smp);
            Rng rng;
            rng.key = rt_rng_key(rp.seed, pixel, smp);
            rng.at(2u + draws_per_node * fld(TF_PRE, i));
            put3(TF_A0, i, shade<M>(sc, h, rng, cnt));
        }
    }
    tree_sync();
    /* 6. colours upwards: R/src/Scene.h:655 (mirror), :705 (glass), a miss is black (:602-605) */
    for (uint32_t l = n_levels; l-- > 0;) {
        const uint32_t b = s_lv[l], e = s_lv[l + 1];
        for (uint32_t i = b + lane; i < e; i += 64) {
            const uint32_t kind = fld(TF_META, i) & 15u;
            if (kind == 1u) continue;
            f3 ret = mk3(0.f, 0.f, 0.f);
            if (kind == 2u) {
                const f3 col = get3(TF_A0, i), r = get3(TF_A0, fld(TF_C1, i));
                const f3 ks = ld3c(sc.materials[ldc(&sc.spheres[fld(TF_OBJ, i)].mat)].ks);
                ret = mk3(col.x + r.x * ks.x, col.y + r.y * ks.y, col.z + r.z * ks.z);
            } else if (kind == 3u) {
                const float F = __uint_as_float(fld(TF_F, i));
                ret = F * get3(TF_A0, fld(TF_C1, i)) + (1.0f - F) * get3(TF_A0, fld(TF_C2, i));
            } else if (kind == 4u) {
                const float F = __uint_as_float(fld(TF_F, i));
                ret = F * get3(TF_A0, fld(TF_C1, i)) + (1.0f - F) * mk3(0.f, 0.f, 0.f);
            }
            put3(TF_A0, i, ret);
        }
        tree_sync();
    }
    if (lane < n_samples) {
        const f3 col = get3(TF_A0, lane);
        float* dst = dl.colors + 3ull * (f_first + lane);
        dst[0] = col.x; dst[1] = col.y; dst[2] = col.z;
    }
    tree_sync(); /* the next batch overwrites the tree */
#ifdef RT_WAVE_TIMES
    if (lane == 0) { s_lv[14] = n_nodes; s_lv[15] = n_list; s_lv[13] = (uint32_t)t_expanded; }
#endif
    return true;
}
#ifdef RT_WAVE_TIMES /* debug build only: start / end clock of every wave of k_render */
RT_DEV uint32_t t_expanded_of(const uint32_t* s_lv) { return s_lv[13]; }
__device__ unsigned long long* rt_dbg_times;
__device__ unsigned long long* rt_dbg_items; /* behind the wave records: one record per worker batch */
extern "C" int rt_debug_wave_times(unsigned long long* d_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(rt_dbg_times), &d_buf, sizeof(d_buf)) == hipSuccess ? 0 : -1;
}
extern "C" int rt_debug_worker_items(unsigned long long* d_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(rt_dbg_items), &d_buf, sizeof(d_buf)) == hipSuccess ? 0 : -1;
}
#endif
template <int M>
RT_DEV uint32_t tree_worker(const DeviceScene& sc, const DCamera* __restrict__ cam, const RenderParams& rp, const DeepList& dl, uint32_t worker,
                            uint32_t spp, uint32_t sample0, uint32_t* s_lv, Cnt& cnt) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t* mem = dl.tree + (size_t)worker * RT_TREE_NODES * RT_TREE_WORDS;
    RenderParams pass = rp; /* the worker handles all samples of the pass, whatever piece of it this launch renders */
    pass.spp = spp; pass.sample0 = sample0;
    uint32_t draws = 0;
    for (uint32_t li = 0; li < sc.n_lights; ++li) draws += ldc(&sc.lights[li].type) == RT_LIGHT_QUAD ? 32u : 0u;
    const uint32_t n_deep = min(dl.head[0], dl.cap);
    const uint64_t n_samples = (uint64_t)n_deep * spp; /* < 2^32: the list's colour table holds them */
    /* Full chunks make a worker's lanes count, small items let the workers end together.  The counter hands out units of
     * half a batch; a worker takes two of them -- a whole batch -- while more units are left than there are workers, and
     * single units after that, so that the last items to start are the short ones (measured with a fixed size, 10 M-triangle
     * share, 2.3 units per worker: halves 25.3 ms, whole batches 26.1 ms; many-spheres kd frame: 41.7 / 38.8 ms). */
    const uint32_t unit = RT_TREE_BATCH / 2;
    const uint64_t n_units = (n_samples + unit - 1) / unit;
    uint32_t samples = 0;
    for (;;) { /* ends for every worker: the counter only grows and n_units is fixed */
        uint32_t first = 0, take = 1;
        if (lane == 0) {
            const uint32_t seen = __hip_atomic_load(&dl.head[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            take = (seen < n_units && n_units - seen > (uint64_t)dl.n_workers) ? 2u : 1u;
            first = atomicAdd(&dl.head[1], take);
        }
        first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        take = (uint32_t)__builtin_amdgcn_readfirstlane((int)take);
        if (first >= n_units) break;
        const uint32_t f_end = (uint32_t)min(n_samples, ((uint64_t)first + take) * unit);
        uint32_t f0 = first * unit, batch = take * unit;
        while (f0 < f_end) {
            const uint32_t n = min(batch, f_end - f0);
            const Cnt before = cnt; /* a batch that does not fit is traced again in halves: its counts are dropped */
#ifdef RT_WAVE_TIMES
            const unsigned long long b_start = wall_clock64();
#endif
            if (tree_batch<M>(sc, cam, pass, mem, s_lv, dl, f0, n, draws, cnt)) {
#ifdef RT_WAVE_TIMES
                if (lane == 0 && rt_dbg_items) { /* one record per batch: start, end, nodes | shaded << 16 | samples << 32 | ticks of the expansion << 40 */
                    unsigned long long* rec = rt_dbg_items + 3ull * (f0 / (RT_TREE_BATCH / 2));
                    rec[0] = b_start; rec[1] = wall_clock64();
                    rec[2] = (unsigned long long)s_lv[14] | ((unsigned long long)s_lv[15] << 16) | ((unsigned long long)n << 32) | ((unsigned long long)(((uint32_t)t_expanded_of(s_lv) - (uint32_t)b_start) & 0xffffffu) << 40); /* ticks until the tree was expanded */
                }
#endif
                f0 += n;
                if (lane == 0) samples += n;
            } else {
                cnt = before;
                batch = max(1u, n / 2);
            }
        }
    }
    return samples;
}

#ifdef RT_WALK_STATS
extern "C" int rt_debug_walk_stats(unsigned long long* out128, int reset) {
    if (reset) { unsigned long long z[128] = {}; return hipMemcpyToSymbol(HIP_SYMBOL(rt_dbg_walk), z, sizeof(z)) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out128, HIP_SYMBOL(rt_dbg_walk), 128 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef RT_LANE_STATS
extern "C" int rt_debug_lane_stats(unsigned long long* out64, int reset) {
    if (reset) { unsigned long long z[64] = {}; return hipMemcpyToSymbol(HIP_SYMBOL(rt_dbg_lanes), z, sizeof(z)) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(rt_dbg_lanes), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef RT_SHADE_HIST
extern "C" int rt_debug_shade_hist(unsigned long long* out65, int reset) {
    if (reset) { unsigned long long z[65] = {}; return hipMemcpyToSymbol(HIP_SYMBOL(rt_dbg_hist), z, sizeof(z)) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out65, HIP_SYMBOL(rt_dbg_hist), 65 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
/* One pass of the render: samples [sample0, sample0 + spp) of every pixel of this rank's tiles.
 * The tile buffer carries image[x + y*w] of R/main.cpp:191 between passes: a pass with sample0 > 0
 * continues the float sum it finds there, in sample order, and the pass that reaches spp_total
 * divides (R/main.cpp:193) -- so k passes give the floats of one pass over all samples.
 * PACK = false: one pixel per wave, lane = sample of a 64-sample chunk (spp > 32).
 * PACK = true : spp <= 32, 64 / spp pixels per wave, lane = (pixel, sample), so that few samples
 *               per pixel do not leave most lanes idle. */
template <int M, bool PACK, bool WORKERS>
__global__ __launch_bounds__(64 * RT_RENDER_WAVES, (M == 0 ? RT_MIN_WAVES0 : RT_MIN_WAVES)) RT_OCC void k_render(DeviceScene sc, const DCamera* __restrict__ cam, RenderParams rp, float* __restrict__ out,
                                                uint64_t* __restrict__ counters, const uint32_t* __restrict__ order, DeepList dl, uint32_t n_workers,
                                                uint32_t pass_spp, uint32_t pass_sample0) {
    const uint32_t lane = threadIdx.x & 63;
    __shared__ uint32_t s_stack[RT_STACK_LDS * 64 * RT_RENDER_WAVES];
    /* WORKERS: the variant that holds the worker code (see RT_RENDER_CASE for which launch takes which) */
    if (WORKERS && blockIdx.x < n_workers) { /* the first workgroups of the launch expand the path trees of the deep list (n_workers > 0 only with RT_RENDER_WAVES == 1) */
        __shared__ uint32_t s_lv[16];
        Cnt wc = new_cnt(s_stack + threadIdx.x, 64 * RT_RENDER_WAVES);
#ifdef RT_WAVE_TIMES
        const unsigned long long w_start = wall_clock64();
#endif
        const uint32_t done = tree_worker<M>(sc, cam, rp, dl, blockIdx.x, pass_spp, pass_sample0, s_lv, wc);
#ifdef RT_WAVE_TIMES
        if (lane == 0 && rt_dbg_times) { /* worker b: after the records of the pixel waves; "pixel" = samples it traced */
            const unsigned long long at = 3ull * ((unsigned long long)(gridDim.x - n_workers) * RT_RENDER_WAVES + blockIdx.x);
            rt_dbg_times[at] = w_start; rt_dbg_times[at + 1] = wall_clock64();
            rt_dbg_times[at + 2] = (0xfffffull << 44) | ((unsigned long long)min(done, 0xfffffu) << 24);
        }
#endif
        flush_counters<M, RT_RENDER_WAVES == 1>(counters, wc, done);
        return;
    }
    const uint8_t* __restrict__ deep_flags = dl.flags;
    /* a single-wave workgroup: the wave number, and with it the tile, the pixel and the output address of an unpacked wave,
     * depend on the workgroup id alone, which lets the compiler keep them in scalar registers */
    const uint32_t wave = (blockIdx.x - n_workers) * (uint32_t)RT_RENDER_WAVES + (RT_RENDER_WAVES == 1 ? 0u : threadIdx.x >> 6);
    const uint32_t px_per_tile = rp.tile_w * rp.tile_h;
    const uint32_t px_per_wave = PACK ? 64u / rp.spp : 1u;
    const uint32_t waves_per_tile = PACK ? (px_per_tile + px_per_wave - 1) / px_per_wave : px_per_tile;
    const uint32_t slot = wave / waves_per_tile; /* position in the launch order */
    const bool wave_has_tile = slot < rp.n_local_tiles; /* wave-uniform */
    const uint32_t local_tile = (order && wave_has_tile) ? ldc(&order[slot]) : slot;
#ifdef RT_WAVE_TIMES
    const unsigned long long t_start = wall_clock64();
#endif
    const uint32_t sub = PACK ? lane / rp.spp : 0u;           /* pixel of the wave this lane works for */
    const uint32_t in_tile = (wave - slot * waves_per_tile) * px_per_wave + sub;
    bool has_pixel = wave_has_tile && sub < px_per_wave && in_tile < px_per_tile;
    /* a pixel on the deep list is left to k_render_wide / k_resolve_deep */
    if (deep_flags && has_pixel && deep_flags[(uint64_t)local_tile * px_per_tile + in_tile]) has_pixel = false;
    const uint32_t tile = rp.rank + local_tile * rp.world;
    const uint32_t x = rt_tile_column(tile, rp.tiles_x) * rp.tile_w + in_tile % rp.tile_w;
    const uint32_t y = (tile / rp.tiles_x) * rp.tile_h + in_tile / rp.tile_w;
    float* dst = out + 3ull * ((uint64_t)local_tile * px_per_tile + in_tile);
    const bool in_image = has_pixel && x < rp.w && y < rp.h; /* padding pixels of edge tiles stay 0 */
    /* path frames: the first L levels in LDS.  A variant without a BVH never touches s_stack (the compiler drops it), so at
     * 4 waves / SIMD = 16 waves per CU each wave can have 10 KB of the CU's 160 KB; the variants that traverse a BVH
     * need their LDS for the traversal stack and keep the frames in the per-lane array */
    constexpr int L = (M & RT_M_BVH) ? 0 : (M == 0 ? RT_FRAME_LDS0 : RT_FRAME_LDS);
    constexpr int SB = M == 0 ? RT_SB0 : (M == RT_M_BVH ? RT_SB2 : RT_SB);
    __shared__ float s_frames[L ? L * RT_FRAME_WORDS * 64 * RT_RENDER_WAVES : 1];
    Cnt cnt = new_cnt(s_stack + threadIdx.x, 64 * RT_RENDER_WAVES, s_frames + (L ? threadIdx.x : 0), 64 * RT_RENDER_WAVES);
    uint32_t my_samples = 0;
    f3 acc = mk3(0.f, 0.f, 0.f); /* identical in every lane of a pixel; the sum a later pass continues is fetched where the
                                  * additions start, not here: it would occupy three registers while the paths are traced */
    if (PACK) {
        const uint32_t first_lane = sub * rp.spp, s = rp.sample0 + (lane - first_lane);
        f3 col = mk3(0.f, 0.f, 0.f);
        if (in_image) {
            Rng rng; rng.key = rt_rng_key(rp.seed, x + y * rp.w, s); rng.at(0);
            float u = ((float)(int)x + rng.unit()) / (float)(int)rp.w;
            float v = ((float)(int)y + rng.unit()) / (float)(int)rp.h;
            f3 o, d;
            primary_ray(*cam, u, v, o, d);
            col = trace_path<M, false, L, SB>(sc, o, norm3_unit(d), rng, cnt);
            ++my_samples;
        }
        if (in_image && rp.sample0 > 0) acc = ld3(dst);
        for (uint32_t i = 0; i < rp.spp; ++i) { /* every lane of the wave takes part in the shuffles */
            const int src = (int)(min(first_lane, 63u) + i) & 63;
            acc.x += __shfl(col.x, src, 64);
            acc.y += __shfl(col.y, src, 64);
            acc.z += __shfl(col.z, src, 64);
        }
    } else {
        for (uint32_t s0 = 0; in_image && s0 < rp.spp; s0 += 64) { /* in_image is wave-uniform here */
            const uint32_t s = rp.sample0 + s0 + lane;
            f3 col = mk3(0.f, 0.f, 0.f);
            if (s0 + lane < rp.spp) {
                Rng rng; rng.key = rt_rng_key(rp.seed, x + y * rp.w, s); rng.at(0);
                /* R/main.cpp:186-187 */
                float u = ((float)(int)x + rng.unit()) / (float)(int)rp.w;
                float v = ((float)(int)y + rng.unit()) / (float)(int)rp.h;
                f3 o, d;
                primary_ray(*cam, u, v, o, d);
                col = trace_path<M, false, L, SB>(sc, o, norm3_unit(d), rng, cnt); /* Ray(pos, dir) normalises again */
                ++my_samples;
            }
            /* image[x + y*w] += color, one sample after the other */
            if (s0 == 0 && rp.sample0 > 0) acc = ld3(dst);
            const uint32_t n = min(64u, rp.spp - s0);
            if (n == 64u) { /* a full chunk: straight-line code, no branch per sample */
#pragma unroll
                for (int i = 0; i < 64; ++i) {
                    acc.x += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(col.x), i));
                    acc.y += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(col.y), i));
                    acc.z += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(col.z), i));
                }
            } else
                for (uint32_t i = 0; i < n; ++i) {
                    acc.x += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(col.x), (int)i));
                    acc.y += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(col.y), (int)i));
                    acc.z += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(col.z), (int)i));
                }
        }
    }
    if (rp.sample0 + rp.spp == rp.spp_total) acc = acc / (float)rp.spp_total; /* image /= nsamples */
    if (has_pixel && lane == (PACK ? sub * rp.spp : 0u)) { dst[0] = in_image ? acc.x : 0.f; dst[1] = in_image ? acc.y : 0.f; dst[2] = in_image ? acc.z : 0.f; }
#ifdef RT_WAVE_TIMES
    {
        const unsigned long long wn = wave_sum(cnt.nodes), wb = wave_sum(cnt.bvh_nodes); /* of the wave: path nodes, BVH nodes fetched */
        if (lane == 0 && rt_dbg_times) {
            rt_dbg_times[3ull * wave] = t_start; rt_dbg_times[3ull * wave + 1] = wall_clock64();
            rt_dbg_times[3ull * wave + 2] = (min(wb >> 6, 0xfffffull) << 44) | (min(wn, 0xfffffull) << 24) | ((unsigned long long)y << 12) | x;
        }
    }
#endif
    flush_counters<M, RT_RENDER_WAVES == 1>(counters, cnt, my_samples); /* every thread of the workgroup reaches this point */
}
/* image[x + y*w] += colour, sample after sample (R/main.cpp:191), for the pixels of the deep list; the pass that
 * reaches spp_total divides (R/main.cpp:193) -- what k_render does for the other pixels.  One thread per pixel. */
__global__ __launch_bounds__(256) void k_resolve_deep(RenderParams rp, DeepList dl, float* __restrict__ out) {
    const uint32_t entry = blockIdx.x * 256u + threadIdx.x;
    if (entry >= min(dl.head[0], dl.cap)) return;
    float* dst = out + 3ull * dl.pixels[entry];
    f3 acc = mk3(0.f, 0.f, 0.f);
    if (rp.sample0 > 0) acc = ld3(dst);
    const float* c = dl.colors + 3ull * (uint64_t)entry * rp.spp;
    for (uint32_t s = 0; s < rp.spp; ++s) { acc.x += c[3 * s]; acc.y += c[3 * s + 1]; acc.z += c[3 * s + 2]; }
    if (rp.sample0 + rp.spp == rp.spp_total) acc = acc / (float)rp.spp_total;
    dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z;
}

/* one thread per pixel: find the pixel in the rank-major tile buffers, quantise like R/main.cpp:213 */
__global__ __launch_bounds__(256) void k_tiles_to_rgb8(const float* __restrict__ tiles, uint32_t w, uint32_t h, uint32_t tile_w, uint32_t tile_h,
                                                       uint32_t tiles_x, uint32_t world, uint32_t per_rank, uint8_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint64_t)w * h) return;
    const uint32_t x = (uint32_t)(i % w), y = (uint32_t)(i / w);
    const uint32_t ty = y / tile_h, tx = x / tile_w;
    const uint32_t t = ty * tiles_x + (tx + tiles_x - ty % tiles_x) % tiles_x; /* inverse of rt_tile_column */
    const uint64_t px = ((uint64_t)(t % world) * per_rank + t / world) * tile_w * tile_h + (uint64_t)(y % tile_h) * tile_w + x % tile_w;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[3 * i + c] = (uint8_t)(int)(255.f * min_std(1.f, tiles[3 * px + c]));
}

/* the same pixel lookup without the quantisation: rank-major tile buffers -> the row-major float image of R/main.cpp:191.
 * One thread per float: consecutive threads write consecutive words. */
__global__ __launch_bounds__(256) void k_tiles_to_image(const float* __restrict__ tiles, uint32_t w, uint32_t h, uint32_t tile_w, uint32_t tile_h,
                                                        uint32_t tiles_x, uint32_t world, uint32_t per_rank, float* __restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (j >= 3ull * w * h) return;
    const uint64_t i = j / 3u;
    const uint32_t c = (uint32_t)(j - 3u * i);
    const uint32_t x = (uint32_t)(i % w), y = (uint32_t)(i / w);
    const uint32_t ty = y / tile_h, tx = x / tile_w;
    const uint32_t t = ty * tiles_x + (tx + tiles_x - ty % tiles_x) % tiles_x; /* inverse of rt_tile_column */
    const uint64_t px = ((uint64_t)(t % world) * per_rank + t / world) * tile_w * tile_h + (uint64_t)(y % tile_h) * tile_w + x % tile_w;
    out[j] = tiles[3 * px + c];
}

RT_DEV rt_hit pack_hit(const FullHit& h, int type, int object, int tri, float u, float v) {
    rt_hit r;
    r.hit = h.hit ? 1 : 0;
    r.type = h.hit ? type : -1;
    r.object = h.hit ? object : -1;
    r.tri = (h.hit && type == RT_MESH) ? tri : -1;
    r.t = h.hit ? h.t : 3.402823466e+38f;
    r.u = (h.hit && type == RT_MESH) ? u : 0.f;
    r.v = (h.hit && type == RT_MESH) ? v : 0.f;
    r.pad0_ = r.pad1_ = r.pad2_ = 0.f;
    r.p[0] = h.p.x; r.p[1] = h.p.y; r.p[2] = h.p.z;
    r.n[0] = h.n.x; r.n[1] = h.n.y; r.n[2] = h.n.z;
    return r;
}

template <int M>
__global__ __launch_bounds__(256) void k_intersect(DeviceScene sc, const float* __restrict__ org, const float* __restrict__ dir,
                                                   size_t n, rt_hit* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    __shared__ uint32_t s_stack[RT_STACK_LDS * RT_STACK_STRIDE];
    f3 o = ld3(org + 3 * i), d = ld3(dir + 3 * i); /* a Ray's direction() is already unit length */
    Cnt cnt = new_cnt(s_stack + threadIdx.x, RT_STACK_STRIDE);
    HitRec rec = closest_rec<M>(sc, o, d, cnt);
    FullHit h = finish_hit(sc, rec, o, d);
    int tri = (h.hit && h.type == RT_MESH) ? (int)ldc(&sc.tri_orig[h.tri]) : -1;
    out[i] = pack_hit(h, h.type, h.obj, tri, rec.u, rec.v);
}

__global__ __launch_bounds__(256) void k_mesh_intersect(DeviceScene sc, uint32_t mesh, const float* __restrict__ org,
                                                        const float* __restrict__ dir, size_t n, rt_hit* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 o = ld3(org + 3 * i), d = ld3(dir + 3 * i); /* a Ray's direction() is already unit length */
    HitRec rec; rec.type = -1; rec.obj = (int)mesh; rec.tri = -1; rec.u = rec.v = 0.f; rec.t = RT_INF;
    __shared__ uint32_t s_stack[RT_STACK_LDS * RT_STACK_STRIDE];
    const DMesh m = ldc(&sc.meshes[mesh]);
    Cnt cnt = new_cnt(s_stack + threadIdx.x, RT_STACK_STRIDE);
    if (mesh_closest<RT_M_BVH>(sc, m, o, d, cnt, rec.t, rec.tri, rec.u, rec.v)) rec.type = RT_MESH;
    FullHit h = finish_hit(sc, rec, o, d);
    out[i] = pack_hit(h, RT_MESH, (int)mesh, h.hit ? (int)ldc(&sc.tri_orig[h.tri]) : -1, rec.u, rec.v);
}

/* Sphere::intersect (R/src/Sphere.h:85-162) / Square::intersect (R/src/Square.h:63-115) of one object alone: no root box,
 * no other object. */
__global__ __launch_bounds__(256) void k_prim_intersect(DeviceScene sc, int type, uint32_t object, const float* __restrict__ org,
                                                       const float* __restrict__ dir, size_t n, rt_hit* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 o = ld3(org + 3 * i), d = ld3(dir + 3 * i); /* a Ray's direction() is already unit length */
    HitRec rec; rec.type = -1; rec.obj = (int)object; rec.tri = -1; rec.u = rec.v = 0.f; rec.t = RT_INF;
    float t;
    if (type == RT_SPHERE ? sphere_t(ldc(&sc.spheres[object]), o, d, t) : square_t(ldc(&sc.squares[object]), o, d, t)) { rec.type = type; rec.t = t; }
    FullHit h = finish_hit(sc, rec, o, d);
    out[i] = pack_hit(h, type, (int)object, -1, 0.f, 0.f);
}

/* Waves per SIMD, shadow-ray batch and LDS path-frame levels of k_trace, per scene variant (lane = ray).  Variant 0 is the
 * one with a choice to make; the table it was taken from is printed by tools/ktrace_table.sh. */
#ifndef RT_KTRACE_WAVES
#define RT_KTRACE_WAVES 4
#endif
#ifndef RT_KTRACE_WAVES0
#define RT_KTRACE_WAVES0 5
#endif
#ifndef RT_KTRACE_SB0
#define RT_KTRACE_SB0 2
#endif
#ifndef RT_KTRACE_FRAMES0
#define RT_KTRACE_FRAMES0 4
#endif
template <int M>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(M == 0 ? RT_KTRACE_WAVES0 : RT_KTRACE_WAVES, M == 0 ? RT_KTRACE_WAVES0 : RT_KTRACE_WAVES))) void k_trace(DeviceScene sc, const float* __restrict__ org, const float* __restrict__ dir, size_t n,
                                               const uint32_t* __restrict__ pixel, const uint32_t* __restrict__ sample, uint32_t seed,
                                               uint32_t k0, float* __restrict__ rgb, uint64_t* __restrict__ counters) {
    __shared__ uint32_t s_stack[RT_STACK_LDS * RT_STACK_STRIDE];
    /* the first levels of the path-frame stack in LDS where no BVH stack needs it */
    constexpr int L = (M & RT_M_BVH) ? 0 : (M == 0 ? RT_KTRACE_FRAMES0 : RT_FRAME_LDS);
    constexpr int SB = M == 0 ? RT_KTRACE_SB0 : (M == RT_M_BVH ? RT_SB2 : RT_SB);
    __shared__ float s_frames[L ? L * RT_FRAME_WORDS * 256 : 1];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Cnt cnt = new_cnt(s_stack + threadIdx.x, RT_STACK_STRIDE, s_frames + (L ? threadIdx.x : 0), 256);
    uint32_t smp = 0;
    if (i < n) {
        f3 o = ld3(org + 3 * i), d = ld3(dir + 3 * i); /* a Ray's direction() is already unit length */
        Rng rng; rng.key = rt_rng_key(seed, pixel[i], sample[i]); rng.at(k0);
        f3 c = trace_path<M, false, L, SB>(sc, o, d, rng, cnt);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
        smp = 1;
    }
    flush_counters<M>(counters, cnt, smp);
}

__global__ __launch_bounds__(256) void k_generate(DCamera cam, const float* __restrict__ u, const float* __restrict__ v, size_t n,
                                                  float* __restrict__ org, float* __restrict__ dir) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 o, d;
    primary_ray(cam, u[i], v[i], o, d);
    org[3 * i] = o.x; org[3 * i + 1] = o.y; org[3 * i + 2] = o.z;
    dir[3 * i] = d.x; dir[3 * i + 1] = d.y; dir[3 * i + 2] = d.z;
}

/* the inverted camera travels as a kernel argument: no host-to-device copy has to be staged */
__global__ __launch_bounds__(64) void k_put_camera(DCamera cam, DCamera* __restrict__ dst) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = cam;
}

/* normalisation through the shortened sequences vs through the IEEE operators, on random vectors */
__global__ __launch_bounds__(256) void k_selftest_arith(uint64_t n, uint32_t seed, unsigned long long* mismatches) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t key = rt_rng_key(seed, (uint32_t)i, (uint32_t)(i >> 32));
    float c[3];
    for (int a = 0; a < 3; ++a) {
        uint32_t m = rt_mix32(key + 2 * a), e = rt_mix32(key + 2 * a + 1);
        uint32_t bits = (m & 0x807fffffu) | ((127u - 44u + e % 89u) << 23); /* sign, mantissa, exponent 2^-44 .. 2^44 */
        if ((e >> 8) % 97u == 0) bits &= 0x80000000u;                           /* now and then an exact zero */
        c[a] = __int_as_float((int)bits);
    }
    f3 v = mk3(c[0], c[1], c[2]);
    float L_fast, L_ref = __builtin_sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    f3 f = norm3_len(v, L_fast);
    f3 r = mk3(v.x / L_ref, v.y / L_ref, v.z / L_ref);
    bool same = __float_as_int(L_fast) == __float_as_int(L_ref) && __float_as_int(f.x) == __float_as_int(r.x) &&
                __float_as_int(f.y) == __float_as_int(r.y) && __float_as_int(f.z) == __float_as_int(r.z);
    bool both_nan = (L_ref != L_ref) && (L_fast != L_fast);
    if (!same && !both_nan) atomicAdd(mismatches, 1ull);

    /* norm3_unit: (a) every sum of squares of its window against the IEEE square root (the first 8193 threads
     * cover the window, the next ones a margin on both sides that must take the ordinary path) */
    if (i < 8193 + 64) {
        float q = __int_as_float(0x3f800000 - 4096 - 32 + (int)i);
        /* a vector whose sum of squares is exactly q: (sqrt-free) one component carries everything */
        float L_bits = __int_as_float((__float_as_int(q) >> 1) + 0x1fc00000);
        bool inside = (uint32_t)(__float_as_int(q) - (0x3f800000 - 4096)) <= 8192u;
        if (inside && __float_as_int(L_bits) != __float_as_int(__builtin_sqrtf(q))) atomicAdd(mismatches, 1ull);
    }
    /* (b) the vector above normalised once with the IEEE operators is unit up to rounding: normalising it
     * again both ways must agree bit for bit (tiny and zero components included) */
    if (L_ref > 0.f && L_ref < 0x1p60f) {
        f3 u1 = r;
        float L2 = __builtin_sqrtf(u1.x * u1.x + u1.y * u1.y + u1.z * u1.z);
        f3 want = mk3(u1.x / L2, u1.y / L2, u1.z / L2), got = norm3_unit(u1);
        bool ok = __float_as_int(want.x) == __float_as_int(got.x) && __float_as_int(want.y) == __float_as_int(got.y) &&
                  __float_as_int(want.z) == __float_as_int(got.z);
        if (!ok && L2 == L2) atomicAdd(mismatches, 1ull);
    }
}


#endif /* RT_KERNELS_H */
