/*
 * rt_api.hip -- host state and the C-ABI entry points of include/rtamd.h; the kernels they launch are in rt_kernels.h.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "rt_build.h"
#include "rt_comm.h"
#include "rt_bvh_device.h"
#include "rt_device.h"
#include "rt_kernels.h"

/* ---- device memory of rt_render_device_noalloc ----------------------------------------------
 * That entry point may not allocate, so what a render needs besides the caller's buffers -- the inverted camera
 * and the workspace of the tile launch order -- lives in this array, which every GPU the library runs on holds
 * for the life of the process.  Host code never uses the array's name as an address: rt_init asks the runtime
 * for the array's address on the current device (hipGetSymbolAddress) and everything goes through that pointer. */
#define RT_NOALLOC_CAM_BYTES 512u
static_assert(sizeof(DCamera) <= RT_NOALLOC_CAM_BYTES, "camera slot of the static scratch array");
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_CAM_BYTES + 2ull * RT_NOALLOC_MAX_TILES * sizeof(uint32_t)];

/* ======================================================================================== */
/*                                       host state                                         */
/* ======================================================================================== */

namespace {

/* launches kernel<mode>: mode = RT_M_KD | RT_M_BVH bits of the scene */
#define RT_DISPATCH(mode, kernel, grid, stream, ...)                                                   \
    do {                                                                                               \
        switch (mode) {                                                                                \
            case 0: hipLaunchKernelGGL(kernel<0>, grid, dim3(256), 0, stream, __VA_ARGS__); break;      \
            case 1: hipLaunchKernelGGL(kernel<1>, grid, dim3(256), 0, stream, __VA_ARGS__); break;      \
            case 2: hipLaunchKernelGGL(kernel<2>, grid, dim3(256), 0, stream, __VA_ARGS__); break;      \
            default: hipLaunchKernelGGL(kernel<3>, grid, dim3(256), 0, stream, __VA_ARGS__); break;     \
        }                                                                                              \
    } while (0)
/* kernel<mode, T> with a workgroup of T threads */
#define RT_DISPATCH_T(mode, kernel, T, grid, stream, ...)                                              \
    do {                                                                                               \
        switch (mode) {                                                                                \
            case 0: hipLaunchKernelGGL((kernel<0, T>), grid, dim3(T), 0, stream, __VA_ARGS__); break;  \
            case 1: hipLaunchKernelGGL((kernel<1, T>), grid, dim3(T), 0, stream, __VA_ARGS__); break;  \
            case 2: hipLaunchKernelGGL((kernel<2, T>), grid, dim3(T), 0, stream, __VA_ARGS__); break;  \
            default: hipLaunchKernelGGL((kernel<3, T>), grid, dim3(T), 0, stream, __VA_ARGS__); break; \
        }                                                                                              \
    } while (0)
/* Every variant exists with and without the worker code, and a launch
 * without workers takes the one without (measured: 100 k-triangle frame 36.5 -> 35.1 ms, 1 M-triangle frame 1 % faster,
 * the Cornell frame unchanged at 20.6 ms -- but 384 instead of 544 bytes of scratch per lane and a third of the spills). */
#define RT_RENDER_CASE(m, p, grid, stream, ...) \
    case (m) * 2 + (p): hipLaunchKernelGGL((k_render<m, (p) != 0, false>), grid, dim3(64 * RT_RENDER_WAVES), 0, stream, __VA_ARGS__); break; \
    case 8 + (m) * 2 + (p): hipLaunchKernelGGL((k_render<m, (p) != 0, true>), grid, dim3(64 * RT_RENDER_WAVES), 0, stream, __VA_ARGS__); break;
#define RT_DISPATCH_RENDER(mode, pack, workers, grid, stream, ...)                                     \
    do {                                                                                               \
        switch (((workers) ? 8 : 0) + ((mode) & 3) * 2 + ((pack) ? 1 : 0)) {                           \
            RT_RENDER_CASE(0, 0, grid, stream, __VA_ARGS__) RT_RENDER_CASE(0, 1, grid, stream, __VA_ARGS__) \
            RT_RENDER_CASE(1, 0, grid, stream, __VA_ARGS__) RT_RENDER_CASE(1, 1, grid, stream, __VA_ARGS__) \
            RT_RENDER_CASE(2, 0, grid, stream, __VA_ARGS__) RT_RENDER_CASE(2, 1, grid, stream, __VA_ARGS__) \
            RT_RENDER_CASE(3, 0, grid, stream, __VA_ARGS__) RT_RENDER_CASE(3, 1, grid, stream, __VA_ARGS__) \
        }                                                                                              \
    } while (0)

struct SceneRes {
    rtb::Built host;
    DeviceScene dev{};
    int mode = 0;
    bool has_bounce = false; /* some sphere is a mirror or glass: only then do path trees differ in depth */
    size_t n_tris = 0, n_bvh = 0, n_bvh4 = 0, n_normals = 0; /* lengths of the device tables (their host copies are dropped) */
    const uint32_t* bvh4_place = nullptr; /* device: where each wide node of the collapse went (treelet order), null = collapse order */
    int bvh_depth = 0;
    rt_build_times times{};
    std::vector<void*> allocs;
    /* for rt_scene_refit: the vertex slice of every mesh, and the box of everything that is not a mesh */
    std::vector<std::pair<uint32_t, uint32_t>> mesh_vertices;
    float static_lo[3] = {INFINITY, INFINITY, INFINITY}, static_hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool leaf_by_count = false; /* the kd root is a leaf whatever the bounds of the objects are */
    bool broken = false;        /* a refit failed half-way: new triangles under old boxes; only rt_scene_destroy takes the handle */
};

const unsigned kCamSlots = 64;
const unsigned kOrderSlots = 8;

struct Lib {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    /* inverted cameras of the renders in flight: a ring in device memory, one slot per rt_render call
     * (the kernel reads its camera through a pointer so the 2 x 16 doubles never occupy registers) */
    DCamera* cam_ring = nullptr;
    unsigned cam_next = 0;
    rt_camera cam_last{};       /* the camera in slot cam_last_slot, uploaded on stream cam_last_stream: a call on the */
    int cam_last_slot = -1;     /* same stream with the same camera does not upload it again                           */
    void* cam_last_stream = nullptr;
    /* workspaces of k_tile_class / k_tile_place, one per render call in turn, grown on demand */
    uint32_t* order_ws[kOrderSlots] = {};   /* keys, order, then the deep list: head, pixels, flags, colours */
    size_t order_cap[kOrderSlots] = {};     /* bytes */
    size_t order_off[kOrderSlots][5] = {};  /* byte offsets of head / pixels / flags / colours / worker trees */
    uint32_t order_workers[kOrderSlots] = {};
    uint32_t order_deep_cap[kOrderSlots] = {};
    size_t order_colors_bytes[kOrderSlots] = {};
    int wide = 1;               /* RTAMD_WIDE: 0 = every path tree depth first, 1 = breadth-first workers where they pay, 2 = wherever something bounces */
    bool treelet_order = false; /* RTAMD_TREELET_ORDER=1 / option "treelet_order": scenes created from now on keep their wide nodes in treelet
                                 * order.  Off by default: frame times and L2 hit rates of the mesh workloads were the same with it (the misses
                                 * are scratch writes and first touches, not neighbours of nodes just fetched), and it costs a second collapse. */
    uint32_t wide_nodes = 12; /* RTAMD_WIDE_NODES: path nodes of the centre ray from which a pixel counts as deep */
    unsigned order_next = 0;
    /* A ring slot is handed out again after kCamSlots / kOrderSlots further calls.  The launches that read it may
     * still be queued on another stream by then, so each slot carries an event recorded behind its last reader and
     * a new user on a different stream waits for it (same stream: the stream's own order suffices). */
    hipEvent_t cam_done[kCamSlots] = {}, order_done[kOrderSlots] = {};
    void* cam_stream[kCamSlots] = {}; void* order_stream[kOrderSlots] = {};
    bool cam_used[kCamSlots] = {}, order_used[kOrderSlots] = {};
    unsigned char* scratch_ptr = nullptr; /* address of the static `scratch` array on this device */
    /* the launch order of the latest probe and what it was made for: the later passes of a progressive render
     * (sample0 > 0, same stream, everything else unchanged) use it again instead of probing the same path trees once more */
    struct OrderKey { rt_scene_t scene; void* stream; rt_camera cam; uint32_t w, h, tile_w, tile_h, rank, world; };
    OrderKey order_key{};
    int order_last_slot = -1;
    bool order_tiles = true;    /* RTAMD_TILE_ORDER=0 renders the tiles in index order */
    /* rt_render_sharded: buffers that live as long as the library does (grown on demand, freed by rt_shutdown), so that a
     * frame loop allocates nothing: this rank's tiles, the gathered tiles and the row-major image (rank 0), the counters, a
     * pinned host block the image is copied into before it reaches the caller's (pageable) buffer, and events around the
     * phases of the latest frame */
    struct Sharded {
        float* mine = nullptr; size_t mine_floats = 0;
        float* all = nullptr; size_t all_floats = 0;
        float* image = nullptr; size_t image_floats = 0;
        float* pinned = nullptr; size_t pinned_floats = 0;
        uint64_t* counters = nullptr;
        hipEvent_t ev[5] = {}; /* start, rendered, gathered, untiled, copied */
        rt_sharded_times last{};
    } sh;
    std::map<rt_scene_t, SceneRes*> scenes;
    rt_scene_t next_handle = 1;
    std::string error;
};
Lib g;

bool g_found_broken = false; /* the latest find() met a scene a failed rt_scene_refit left inconsistent */
int fail(int code, const std::string& msg) {
    g.error = msg;
    if (code == RT_E_HANDLE && g_found_broken) g.error += " (a failed rt_scene_refit left this scene inconsistent: destroy it and create it again)";
    g_found_broken = false;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(RT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));              \
    } while (0)

template <class T>
int upload(SceneRes* r, const std::vector<T>& v, const T** dst) {
    *dst = nullptr;
    if (v.empty()) return RT_OK;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
    r->allocs.push_back(p);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *dst = (const T*)p;
    return RT_OK;
}

void release(SceneRes* r) {
    for (void* p : r->allocs) (void)hipFree(p);
    delete r;
}

SceneRes* find(rt_scene_t h) {
    auto it = g.scenes.find(h);
    g_found_broken = it != g.scenes.end() && it->second->broken;
    return (it == g.scenes.end() || it->second->broken) ? nullptr : it->second;
}

/* device temporaries of the host-pointer entry points */
struct Scratch {
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
    int get(void** out, size_t bytes) {
        *out = nullptr;
        if (bytes == 0) bytes = 16;
        HIP_TRY(hipMalloc(out, bytes));
        ptrs.push_back(*out);
        return RT_OK;
    }
};

/* Device path of rt_scene_create_ex, part 1: triangle records and mesh bounds from the caller's raw vertex / index
 * pools (rt_bvh_device.hip).  Fills r->dev.tris / tri_orig / normals and the bounds of r->host.meshes; the meshes
 * large enough for a BVH come back in bvh_todo with the pad of their node boxes. */
int meshes_on_device(const rt_scene_desc& desc, SceneRes* r, std::vector<uint32_t>& bvh_todo, std::vector<float>& bvh_pad) {
    DeviceScene& d = r->dev;
    const size_t n_mesh = r->host.meshes.size();
    uint64_t total = 0;
    for (const DMesh& m : r->host.meshes) total += m.n_tris;
    Scratch tmp;
    float* d_pos; uint32_t* d_idx; uint32_t* d_box; float* d_nrm = nullptr; DTri* d_tris = nullptr; uint32_t* d_orig = nullptr;
    int rc;
    if ((rc = tmp.get((void**)&d_pos, (size_t)desc.n_vertices * 12)) || (rc = tmp.get((void**)&d_idx, (size_t)desc.n_triangles * 12)) ||
        (rc = tmp.get((void**)&d_box, n_mesh * 32))) return rc;
    hipError_t e = hipSuccess;
    if (desc.n_vertices) {
        e = hipMalloc((void**)&d_nrm, (size_t)desc.n_vertices * 12);
        if (e == hipSuccess) r->allocs.push_back(d_nrm);
    }
    if (e == hipSuccess && total) { e = hipMalloc((void**)&d_tris, (size_t)total * sizeof(DTri)); if (e == hipSuccess) r->allocs.push_back(d_tris); }
    if (e == hipSuccess && total) { e = hipMalloc((void**)&d_orig, (size_t)total * 4); if (e == hipSuccess) r->allocs.push_back(d_orig); }
    std::vector<uint32_t> boxes(n_mesh * 8);
    for (size_t k = 0; k < n_mesh; ++k) rtd::box_init(&boxes[8 * k]);
    if (e == hipSuccess && desc.n_vertices) e = hipMemcpyAsync(d_pos, desc.positions, (size_t)desc.n_vertices * 12, hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess && desc.n_vertices) e = hipMemcpyAsync(d_nrm, desc.normals, (size_t)desc.n_vertices * 12, hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess && desc.n_triangles) e = hipMemcpyAsync(d_idx, desc.triangles, (size_t)desc.n_triangles * 12, hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_box, boxes.data(), n_mesh * 32, hipMemcpyHostToDevice, g.stream);
    for (size_t k = 0; k < n_mesh && e == hipSuccess; ++k) {
        const rt_mesh& m = desc.meshes[k];
        const DMesh& o = r->host.meshes[k];
        e = rtd::make_mesh_tris(d_pos + 3 * (size_t)m.first_vertex, d_idx + 3 * (size_t)m.first_triangle, m.first_vertex, m.n_vertices, m.n_triangles,
                                d_tris + o.first_tri, d_orig + o.first_tri, d_box + 8 * k, g.stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(boxes.data(), d_box, n_mesh * 32, hipMemcpyDeviceToHost, g.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (e != hipSuccess) return fail(RT_E_HIP, std::string("rt_scene_create: triangle records: ") + hipGetErrorString(e));
    for (size_t k = 0; k < n_mesh; ++k) {
        float lo[3], hi[3];
        bool bad = false;
        rtd::box_decode(&boxes[8 * k], lo, hi, bad);
        if (bad) return fail(RT_E_INVALID, "rt_scene_create: triangle vertex index out of range");
        DMesh& o = r->host.meshes[k];
        const float pad = rtb::mesh_bounds_from_box(lo, hi, o.n_tris, o);
        if (o.n_tris > rtb::kBvhMinTris) { bvh_todo.push_back((uint32_t)k); bvh_pad.push_back(pad); }
    }
    d.tris = d_tris; d.tri_orig = d_orig; d.normals = d_nrm;
    r->n_tris = (size_t)total;
    r->n_normals = 3 * (size_t)desc.n_vertices;
    return RT_OK;
}

/* Part 2: one BVH per mesh of bvh_todo, built from the records on the device; the node tables are joined into
 * r->dev.bvh (child indices already carry each mesh's offset). */
int bvhs_on_device(SceneRes* r, const std::vector<uint32_t>& bvh_todo, const std::vector<float>& bvh_pad) {
    DeviceScene& d = r->dev;
    std::vector<rtd::DeviceBvh> parts;
    std::string err;
    int32_t base = 0;
    hipError_t e = hipSuccess;
    for (size_t k = 0; k < bvh_todo.size() && e == hipSuccess; ++k) {
        DMesh& m = r->host.meshes[bvh_todo[k]];
        rtd::DeviceBvh b;
        e = rtd::build_mesh_bvh(const_cast<DTri*>(d.tris), const_cast<uint32_t*>(d.tri_orig), m.first_tri, m.n_tris, m.lo, m.hi,
                                bvh_pad[k], base, rtb::kBvhMaxDepth, g.stream, b, err);
        if (e != hipSuccess) break;
        m.bvh_root = b.root;
        base += (int32_t)b.n_nodes;
        r->times.device_sort_ms += b.sort_ms; r->times.device_tree_ms += b.tree_ms;
        r->bvh_depth = b.max_depth > r->bvh_depth ? b.max_depth : r->bvh_depth;
        parts.push_back(b);
    }
    DBvhNode* all = nullptr;
    if (e == hipSuccess && parts.size() == 1) {
        all = parts[0].nodes; parts.clear();
    } else if (e == hipSuccess) {
        e = hipMalloc((void**)&all, (size_t)base * sizeof(DBvhNode));
        size_t at = 0;
        for (size_t k = 0; k < parts.size() && e == hipSuccess; ++k) {
            e = hipMemcpy(all + at, parts[k].nodes, (size_t)parts[k].n_nodes * sizeof(DBvhNode), hipMemcpyDeviceToDevice);
            at += parts[k].n_nodes;
        }
        if (e != hipSuccess && all) { (void)hipFree(all); all = nullptr; }
        if (e != hipSuccess) err = std::string("rt_scene_create: node table: ") + hipGetErrorString(e);
    }
    for (rtd::DeviceBvh& b : parts) (void)hipFree(b.nodes);
    if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? RT_E_LIMIT : RT_E_HIP, err);
    r->allocs.push_back(all);
    d.bvh = all;
    r->n_bvh = (size_t)base;
    return RT_OK;
}

/* Part 3 (both build paths): the four-wide trees the rays walk, made on the device from the binary node table. */
int bvh4_on_device(SceneRes* r) {
    if (!r->n_bvh) return RT_OK;
    std::vector<int32_t> roots, roots4;
    std::vector<size_t> owner;
    for (size_t k = 0; k < r->host.meshes.size(); ++k)
        if (r->host.meshes[k].bvh_root >= 0) { roots.push_back(r->host.meshes[k].bvh_root); owner.push_back(k); }
    roots4.assign(roots.size(), -1);
    DBvh4Node* nodes = nullptr;
    uint32_t n4 = 0;
    std::string err;
    hipError_t e = rtd::collapse_to_bvh4(r->dev.bvh, (uint32_t)r->n_bvh, roots.data(), (uint32_t)roots.size(), g.stream, &nodes, &n4, roots4.data(), err);
    if (e != hipSuccess) return fail(RT_E_HIP, "rt_scene_create: " + err);
    r->allocs.push_back(nodes);
    r->dev.bvh4 = nodes;
    r->n_bvh4 = n4;
    DBvh4Q* qnodes = nullptr;
    HIP_TRY(hipMalloc((void**)&qnodes, (size_t)(n4 ? n4 : 1) * sizeof(DBvh4Q)));
    r->allocs.push_back(qnodes);
    r->dev.bvh4q = qnodes;
    HIP_TRY(rtd::quantise_bvh4(nodes, n4, qnodes, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    if (g.treelet_order && n4 > 1) { /* the same trees once more, every node written to its place in treelet order */
        uint32_t* place = nullptr;
        e = rtd::treelet_places(qnodes, n4, roots4.data(), (uint32_t)roots4.size(), &place, err);
        if (e != hipSuccess) return fail(RT_E_HIP, "rt_scene_create: " + err);
        r->allocs.push_back(place);
        r->bvh4_place = place;
        DBvh4Node* same = nullptr;
        e = rtd::collapse_to_bvh4(r->dev.bvh, (uint32_t)r->n_bvh, roots.data(), (uint32_t)roots.size(), g.stream, &same, &n4, roots4.data(), err, nodes, n4, place);
        if (e != hipSuccess) return fail(RT_E_HIP, "rt_scene_create: " + err);
        HIP_TRY(rtd::quantise_bvh4(nodes, n4, qnodes, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
    }
    for (size_t k = 0; k < owner.size(); ++k) r->host.meshes[owner[k]].bvh4_root = roots4[k];
    return RT_OK;
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

int make_render_params(uint32_t w, uint32_t h, uint32_t spp, uint32_t seed, uint32_t tile_w, uint32_t tile_h, uint32_t rank,
                       uint32_t world, RenderParams& rp) {
    if (w == 0 || h == 0 || spp == 0 || tile_w == 0 || tile_h == 0 || world == 0 || rank >= world)
        return fail(RT_E_INVALID, "rt_render: zero size or rank >= world");
    if ((uint64_t)w * h > 0x7fffffffull) return fail(RT_E_LIMIT, "rt_render: more than 2^31 pixels");
    rp.w = w; rp.h = h; rp.spp = spp; rp.seed = seed;
    rp.tile_w = tile_w; rp.tile_h = tile_h;
    rp.tiles_x = (w + tile_w - 1) / tile_w;
    uint32_t tiles_y = (h + tile_h - 1) / tile_h;
    rp.n_tiles = rp.tiles_x * tiles_y;
    rp.rank = rank; rp.world = world;
    rp.n_local_tiles = (uint32_t)rt_tiles_per_rank(w, h, tile_w, tile_h, rank, world);
    rp.sample0 = 0; rp.spp_total = spp; rp.pad[0] = rp.pad[1] = rp.pad[2] = 0;
    uint64_t waves = (uint64_t)rp.n_local_tiles * tile_w * tile_h;
    if (waves / RT_RENDER_WAVES + 1 > 0x7fffffffull) return fail(RT_E_LIMIT, "rt_render: grid too large");
    return RT_OK;
}

}  // namespace

/* ======================================================================================== */
/*                                        C-ABI                                             */
/* ======================================================================================== */

extern "C" {

const char* rt_last_error(void) { return g.error.c_str(); }

int rt_init(int device) {
    if (g.ready) {
        if (device == g.device) return RT_OK;
        return fail(RT_E_INVALID, "rt_init: already initialised on another device; call rt_shutdown first");
    }
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(RT_E_INVALID, "rt_init: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&g.ev0));
    HIP_TRY(hipEventCreate(&g.ev1));
    HIP_TRY(hipMalloc((void**)&g.cam_ring, kCamSlots * sizeof(DCamera)));
    for (unsigned i = 0; i < kCamSlots; ++i) { HIP_TRY(hipEventCreateWithFlags(&g.cam_done[i], hipEventDisableTiming)); g.cam_used[i] = false; }
    for (unsigned i = 0; i < kOrderSlots; ++i) { HIP_TRY(hipEventCreateWithFlags(&g.order_done[i], hipEventDisableTiming)); g.order_used[i] = false; }
    /* this device's copy of the static array (each GPU holds its own, at its own address) */
    g.scratch_ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress((void**)&g.scratch_ptr, HIP_SYMBOL(scratch)));
    g.cam_next = 0;
    g.cam_last_slot = -1;
    const char* ord = std::getenv("RTAMD_TILE_ORDER");
    g.order_tiles = !(ord && ord[0] == '0');
    const char* wd = std::getenv("RTAMD_WIDE");
    g.wide = (wd && wd[0] >= '0' && wd[0] <= '2') ? wd[0] - '0' : 1;
    const char* wn = std::getenv("RTAMD_WIDE_NODES");
    g.wide_nodes = 12;
    if (wn && std::atoi(wn) >= 1 && std::atoi(wn) <= RT_PROBE_CAP) g.wide_nodes = (uint32_t)std::atoi(wn);
    const char* to = std::getenv("RTAMD_TREELET_ORDER");
    g.treelet_order = to && to[0] == '1';
    g.device = device;
    g.ready = true;
    return RT_OK;
}

int rt_set_option(const char* name, int64_t value) {
    if (!name) return fail(RT_E_INVALID, "rt_set_option: null name");
    const std::string k(name);
    if (k == "tile_order") g.order_tiles = value != 0;
    else if (k == "treelet_order") g.treelet_order = value != 0;
    else if (k == "wide") {
        if (value < 0 || value > 2) return fail(RT_E_INVALID, "rt_set_option: wide must be 0, 1 or 2");
        g.wide = (int)value;
    }
    else if (k == "wide_nodes") {
        if (value < 1 || value > RT_PROBE_CAP) return fail(RT_E_INVALID, "rt_set_option: wide_nodes must lie in 1 .. 24");
        g.wide_nodes = (uint32_t)value;
    } else return fail(RT_E_INVALID, "rt_set_option: unknown option " + k);
    g.order_last_slot = -1; /* a cached launch order / deep list was made under the old settings */
    return RT_OK;
}

int rt_shutdown(void) {
    if (!g.ready) return RT_OK;
    rtc::destroy();
    (void)hipSetDevice(g.device);
    (void)hipStreamSynchronize(g.stream);
    for (auto& kv : g.scenes) release(kv.second);
    g.scenes.clear();
    (void)hipFree(g.cam_ring);
    g.cam_ring = nullptr;
    for (unsigned i = 0; i < kOrderSlots; ++i) { (void)hipFree(g.order_ws[i]); g.order_ws[i] = nullptr; g.order_cap[i] = 0; }
    g.order_last_slot = -1;
    for (unsigned i = 0; i < kCamSlots; ++i) if (g.cam_done[i]) { (void)hipEventDestroy(g.cam_done[i]); g.cam_done[i] = nullptr; }
    for (unsigned i = 0; i < kOrderSlots; ++i) if (g.order_done[i]) { (void)hipEventDestroy(g.order_done[i]); g.order_done[i] = nullptr; }
    g.scratch_ptr = nullptr;
    (void)hipFree(g.sh.mine); (void)hipFree(g.sh.all); (void)hipFree(g.sh.image); (void)hipFree(g.sh.counters);
    if (g.sh.pinned) (void)hipHostFree(g.sh.pinned);
    for (hipEvent_t& e : g.sh.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    g.sh = Lib::Sharded();
    (void)hipEventDestroy(g.ev0);
    (void)hipEventDestroy(g.ev1);
    (void)hipStreamDestroy(g.stream);
    g.stream = nullptr; g.ev0 = g.ev1 = nullptr;
    g.ready = false; g.device = -1;
    return RT_OK;
}

int rt_selftest_arith(uint64_t n, uint32_t seed, uint64_t* mismatches) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_selftest_arith: call rt_init first");
    if (!mismatches) return fail(RT_E_INVALID, "rt_selftest_arith: null output");
    if (n > (1ull << 38)) return fail(RT_E_LIMIT, "rt_selftest_arith: n too large");
    *mismatches = 0;
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(g.device));
    Scratch tmp;
    void* d_m;
    int rc;
    if ((rc = tmp.get(&d_m, 8))) return rc;
    HIP_TRY(hipMemsetAsync(d_m, 0, 8, g.stream));
    hipLaunchKernelGGL(k_selftest_arith, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.stream, n, seed, (unsigned long long*)d_m);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(mismatches, d_m, 8, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return RT_OK;
}

int rt_scene_check(const rt_scene_desc* desc, rt_build_info* info) {
    if (!desc || !info) return fail(RT_E_INVALID, "rt_scene_check: null argument");
    std::memset(info, 0, sizeof(*info));
    if (desc->n_triangles >= RT_MAX_TRIANGLES) return fail(RT_E_LIMIT, "rt_scene_check: 2^28 triangles or more (BVH leaf entries hold a 28-bit triangle position)");
    rtb::Built b;
    std::string err;
    if (!rtb::build_scene(*desc, b, err)) return fail(RT_E_INVALID, err);
    int kd_depth = 0, bvh_depth = 0;
    for (const DKdNode& k : b.kd) kd_depth = k.depth > kd_depth ? k.depth : kd_depth;
    if (kd_depth >= RT_KD_MAX_DEPTH) return fail(RT_E_LIMIT, "rt_scene_check: kd-tree deeper than 15 levels");
    if (!rtb::verify_bvh(b, bvh_depth, err)) return fail(RT_E_LIMIT, err);
    info->kd_nodes = (uint32_t)b.kd.size(); info->kd_max_depth = (uint32_t)kd_depth;
    for (const DMesh& m : b.meshes) info->bvh_meshes += m.bvh_root >= 0 ? 1u : 0u;
    info->bvh_nodes = (uint32_t)b.bvh.size(); info->bvh_max_depth = (uint32_t)bvh_depth;
    info->kernel_variant = ((!b.kd.empty() && b.kd[0].leaf) ? 0u : (uint32_t)RT_M_KD) | (b.bvh.empty() ? 0u : (uint32_t)RT_M_BVH);
    info->device_bytes = b.spheres.size() * sizeof(DSphere) + b.squares.size() * sizeof(DSquare) + b.meshes.size() * sizeof(DMesh) +
                         b.tris.size() * sizeof(DTri) + b.tri_orig.size() * 4 + b.bvh.size() * sizeof(DBvhNode) + b.normals.size() * 4 +
                         b.materials.size() * sizeof(DMaterial) + b.lights.size() * sizeof(DLight) + b.kd.size() * sizeof(DKdNode) +
                         b.refs.size() * sizeof(DRef);
    return RT_OK;
}

int rt_scene_create_ex(const rt_scene_desc* desc, uint32_t flags, rt_scene_t* out) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_scene_create: call rt_init first");
    if (!desc || !out) return fail(RT_E_INVALID, "rt_scene_create: null argument");
    if (flags & ~(uint32_t)RT_BUILD_BVH_DEVICE) return fail(RT_E_INVALID, "rt_scene_create_ex: unknown flag");
    if (desc->n_triangles >= RT_MAX_TRIANGLES) return fail(RT_E_LIMIT, "rt_scene_create: 2^28 triangles or more (BVH leaf entries hold a 28-bit triangle position)");
    HIP_TRY(hipSetDevice(g.device));
    const bool device_bvh = (flags & RT_BUILD_BVH_DEVICE) != 0;
    const auto t0 = std::chrono::steady_clock::now();
    SceneRes* r = new SceneRes();
    std::string err;
    if (!rtb::build_scene(*desc, r->host, err, device_bvh)) { delete r; return fail(RT_E_INVALID, err); }
    int max_depth = 0;
    for (const DKdNode& k : r->host.kd) max_depth = k.depth > max_depth ? k.depth : max_depth;
    if (max_depth >= RT_KD_MAX_DEPTH) { delete r; return fail(RT_E_LIMIT, "rt_scene_create: kd-tree deeper than 15 levels"); }
    r->times.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    DeviceScene& d = r->dev;
    int rc;
    if ((rc = upload(r, r->host.spheres, &d.spheres)) || (rc = upload(r, r->host.squares, &d.squares)) ||
        (rc = upload(r, r->host.tris, &d.tris)) || (rc = upload(r, r->host.tri_orig, &d.tri_orig)) ||
        (rc = upload(r, r->host.bvh, &d.bvh)) ||
        (rc = upload(r, r->host.normals, &d.normals)) || (rc = upload(r, r->host.materials, &d.materials)) ||
        (rc = upload(r, r->host.lights, &d.lights)) || (rc = upload(r, r->host.kd, &d.kd)) ||
        (rc = upload(r, r->host.refs, &d.refs)) || (rc = upload(r, r->host.leaf_recs, &d.leaf_recs)) ||
        (rc = upload(r, r->host.light_sq, &d.light_sq))) {
        release(r);
        return rc;
    }
    r->n_tris = r->host.tris.size();
    r->n_normals = r->host.normals.size();
    r->n_bvh = r->host.bvh.size();
    std::vector<uint32_t> bvh_todo; /* device path: meshes that get a BVH, and the pad of their node boxes */
    std::vector<float> bvh_pad;
    if (device_bvh && !r->host.meshes.empty() && (rc = meshes_on_device(*desc, r, bvh_todo, bvh_pad))) { release(r); return rc; }
    if (!bvh_todo.empty() && (rc = bvhs_on_device(r, bvh_todo, bvh_pad))) { release(r); return rc; }
    if ((rc = bvh4_on_device(r))) { release(r); return rc; }
    if ((rc = upload(r, r->host.meshes, &d.meshes))) { release(r); return rc; }
    d.n_spheres = (uint32_t)r->host.spheres.size(); d.n_squares = (uint32_t)r->host.squares.size();
    d.n_meshes = (uint32_t)r->host.meshes.size(); d.n_lights = (uint32_t)r->host.lights.size();
    d.n_kd = (uint32_t)r->host.kd.size(); d.n_refs = (uint32_t)r->host.refs.size();
    for (int a = 0; a < 3; ++a) { d.box_lo[a] = r->host.box_lo[a]; d.box_hi[a] = r->host.box_hi[a]; }
    d.kd_epsilon = r->host.kd_epsilon;
    d.cert_radius = r->host.cert_radius;
    d.root_is_leaf = (!r->host.kd.empty() && r->host.kd[0].leaf) ? 1 : 0;
    r->mode = (d.root_is_leaf ? 0 : RT_M_KD) | (r->n_bvh ? RT_M_BVH : 0);
    for (const DSphere& sp : r->host.spheres) {
        const int mt = r->host.materials[(size_t)sp.mat].type;
        r->has_bounce = r->has_bounce || mt == RT_MAT_MIRROR || mt == RT_MAT_GLASS;
    }
    for (uint32_t i = 0; i < desc->n_meshes; ++i) r->mesh_vertices.push_back({desc->meshes[i].first_vertex, desc->meshes[i].n_vertices});
    r->leaf_by_count = (int64_t)r->host.refs.size() <= (int64_t)desc->kd_max_leaf || desc->kd_max_depth <= 0;
    { /* Scene::computeGlobalAABB's contributions of squares and spheres (R/src/Scene.h:133-166), as rtb::build_scene adds them */
        auto grow = [&](const float* p) {
            for (int a = 0; a < 3; ++a) { r->static_lo[a] = (p[a] < r->static_lo[a]) ? p[a] : r->static_lo[a]; r->static_hi[a] = (r->static_hi[a] < p[a]) ? p[a] : r->static_hi[a]; }
        };
        for (uint32_t i = 0; i < desc->n_squares; ++i)
            for (int v = 0; v < 4; ++v) grow(desc->squares[i].v[v]);
        for (uint32_t i = 0; i < desc->n_spheres; ++i) {
            const rt_sphere& sp = desc->spheres[i];
            const float lo[3] = {sp.center[0] - sp.radius, sp.center[1] - sp.radius, sp.center[2] - sp.radius};
            const float hi[3] = {sp.center[0] + sp.radius, sp.center[1] + sp.radius, sp.center[2] + sp.radius};
            grow(lo); grow(hi);
        }
    }
    /* the big tables now live on the device only (rt_scene_verify reads them back) */
    std::vector<DTri>().swap(r->host.tris);
    std::vector<uint32_t>().swap(r->host.tri_orig);
    std::vector<DBvhNode>().swap(r->host.bvh);
    std::vector<float>().swap(r->host.normals);
    r->times.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    rt_scene_t h = g.next_handle++;
    g.scenes[h] = r;
    *out = h;
    /* RTAMD_VERIFY_BUILD=1: read the tables back and run the host verifier on them before the scene is handed out
     * (every triangle in one leaf, boxes nested, depth within the traversal stack) -- what rt_scene_verify does on request */
    const char* vb = std::getenv("RTAMD_VERIFY_BUILD");
    if (vb && vb[0] == '1' && r->n_bvh) {
        const int rc_verify = rt_scene_verify(h, nullptr, nullptr);
        if (rc_verify) { const std::string why = g.error; (void)rt_scene_destroy(h); *out = 0; return fail(rc_verify, "rt_scene_create: built tree rejected: " + why); }
    }
    return RT_OK;
}

int rt_scene_create(const rt_scene_desc* desc, rt_scene_t* out) {
    /* host builder (the better tree) unless the meshes are so large that its seconds would dominate */
    const char* b = std::getenv("RTAMD_BVH_BUILD");
    bool device = desc && desc->n_triangles >= RT_BUILD_AUTO_DEVICE_TRIANGLES;
    if (b && std::strcmp(b, "device") == 0) device = true;
    if (b && std::strcmp(b, "host") == 0) device = false;
    return rt_scene_create_ex(desc, device ? (uint32_t)RT_BUILD_BVH_DEVICE : 0u, out);
}

int rt_scene_verify(rt_scene_t scene, rt_build_info* info, rt_build_times* times) {
    SceneRes* r = find(scene);
    if (!r) return fail(RT_E_HANDLE, "rt_scene_verify: unknown scene");
    HIP_TRY(hipSetDevice(g.device));
    HIP_TRY(hipStreamSynchronize(g.stream));
    rtb::Built b; /* the tables the kernels read, copied back */
    b.meshes = r->host.meshes;
    b.tris.resize(r->n_tris); b.tri_orig.resize(r->n_tris); b.bvh.resize(r->n_bvh);
    if (r->n_tris) {
        HIP_TRY(hipMemcpy(b.tris.data(), r->dev.tris, r->n_tris * sizeof(DTri), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(b.tri_orig.data(), r->dev.tri_orig, r->n_tris * 4, hipMemcpyDeviceToHost));
    }
    if (r->n_bvh) HIP_TRY(hipMemcpy(b.bvh.data(), r->dev.bvh, r->n_bvh * sizeof(DBvhNode), hipMemcpyDeviceToHost));
    int depth = 0, stack4 = 0;
    std::string err;
    if (!rtb::verify_bvh(b, depth, err)) return fail(RT_E_LIMIT, err);
    std::vector<DBvh4Node> wide(r->n_bvh4); /* and the four-wide trees the rays actually walk */
    if (r->n_bvh4) HIP_TRY(hipMemcpy(wide.data(), r->dev.bvh4, r->n_bvh4 * sizeof(DBvh4Node), hipMemcpyDeviceToHost));
    if (!rtb::verify_bvh4(b, wide, stack4, err)) return fail(RT_E_LIMIT, err);
    std::vector<DBvh4Q> quant(r->n_bvh4); /* ... in the 64-byte form they are fetched in */
    if (r->n_bvh4) HIP_TRY(hipMemcpy(quant.data(), r->dev.bvh4q, r->n_bvh4 * sizeof(DBvh4Q), hipMemcpyDeviceToHost));
    if (!rtb::verify_bvh4q(wide, quant, err)) return fail(RT_E_LIMIT, err);
    if (info) {
        std::memset(info, 0, sizeof(*info));
        int kd_depth = 0;
        for (const DKdNode& k : r->host.kd) kd_depth = k.depth > kd_depth ? k.depth : kd_depth;
        info->kd_nodes = (uint32_t)r->host.kd.size(); info->kd_max_depth = (uint32_t)kd_depth;
        for (const DMesh& m : b.meshes) info->bvh_meshes += m.bvh_root >= 0 ? 1u : 0u;
        info->bvh_nodes = (uint32_t)r->n_bvh; info->bvh_max_depth = (uint32_t)depth;
        info->kernel_variant = (uint32_t)r->mode;
        info->device_bytes = r->host.spheres.size() * sizeof(DSphere) + r->host.squares.size() * sizeof(DSquare) + b.meshes.size() * sizeof(DMesh) +
                             r->n_tris * (sizeof(DTri) + 4) + r->n_bvh * sizeof(DBvhNode) + r->n_bvh4 * (sizeof(DBvh4Node) + sizeof(DBvh4Q)) + (size_t)r->n_normals * 4 +
                             r->host.materials.size() * sizeof(DMaterial) + r->host.lights.size() * sizeof(DLight) +
                             r->host.kd.size() * sizeof(DKdNode) + r->host.refs.size() * sizeof(DRef);
    }
    if (times) *times = r->times;
    return RT_OK;
}

int rt_scene_refit(rt_scene_t scene, const float* positions, const float* normals, uint32_t n_vertices, float* device_ms) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_scene_refit: call rt_init first");
    SceneRes* r = find(scene);
    if (!r) return fail(RT_E_HANDLE, "rt_scene_refit: unknown scene");
    if (!positions || !normals) return fail(RT_E_INVALID, "rt_scene_refit: null argument");
    if ((size_t)n_vertices * 3 != r->n_normals) return fail(RT_E_INVALID, "rt_scene_refit: the scene was made with another number of vertices");
    if (!r->dev.root_is_leaf || !r->leaf_by_count)
        return fail(RT_E_INVALID, "rt_scene_refit: the object kd-tree of this scene depends on the mesh bounds (more objects than a leaf holds): make the scene again");
    if (device_ms) *device_ms = 0.f;
    if (n_vertices == 0 || r->host.meshes.empty()) return RT_OK;
    HIP_TRY(hipSetDevice(g.device));
    HIP_TRY(hipStreamSynchronize(g.stream));
    const size_t n_mesh = r->host.meshes.size();
    Scratch tmp;
    float* d_pos; uint32_t* d_box;
    int rc;
    if ((rc = tmp.get((void**)&d_pos, (size_t)n_vertices * 12)) || (rc = tmp.get((void**)&d_box, n_mesh * 64))) return rc;
    std::vector<uint32_t> boxes(n_mesh * 16); /* per mesh: the box of its records' corners, the box of its raw vertices */
    for (size_t k = 0; k < 2 * n_mesh; ++k) rtd::box_init(&boxes[8 * k]);
    HIP_TRY(hipEventRecord(g.ev0, g.stream));
    HIP_TRY(hipMemcpyAsync(d_pos, positions, (size_t)n_vertices * 12, hipMemcpyHostToDevice, g.stream));
    /* From here on the scene's tables are overwritten in place.  An error return before the end leaves new triangles under
     * old boxes, which later queries would silently miss: the handle is then marked and refused until it is destroyed. */
    struct Guard { SceneRes* r; bool done; ~Guard() { if (!done) r->broken = true; } } guard{r, false};
    HIP_TRY(hipMemcpyAsync(const_cast<float*>(r->dev.normals), normals, (size_t)n_vertices * 12, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(d_box, boxes.data(), n_mesh * 64, hipMemcpyHostToDevice, g.stream));
    for (size_t k = 0; k < n_mesh; ++k) {
        const DMesh& m = r->host.meshes[k];
        HIP_TRY(rtd::remake_mesh_tris(d_pos, n_vertices, const_cast<DTri*>(r->dev.tris) + m.first_tri, m.n_tris, d_box + 16 * k, g.stream));
        HIP_TRY(rtd::vertex_box(d_pos + 3 * (size_t)r->mesh_vertices[k].first, r->mesh_vertices[k].second, d_box + 16 * k + 8, g.stream));
    }
    HIP_TRY(hipMemcpyAsync(boxes.data(), d_box, n_mesh * 64, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    /* bounds of every mesh and of the scene, as a fresh build computes them */
    float root_lo[3], root_hi[3];
    for (int a = 0; a < 3; ++a) { root_lo[a] = r->static_lo[a]; root_hi[a] = r->static_hi[a]; }
    std::vector<int32_t> roots;
    std::vector<float> pads;
    for (size_t k = 0; k < n_mesh; ++k) {
        float lo[3], hi[3];
        bool bad = false;
        rtd::box_decode(&boxes[16 * k], lo, hi, bad);
        if (bad) return fail(RT_E_INVALID, "rt_scene_refit: a triangle record holds a vertex id outside the vertex pool");
        DMesh& m = r->host.meshes[k];
        const float pad = rtb::mesh_bounds_from_box(lo, hi, m.n_tris, m);
        if (m.bvh_root >= 0) { roots.push_back(m.bvh_root); pads.push_back(pad); }
        rtd::box_decode(&boxes[16 * k + 8], lo, hi, bad);
        for (int a = 0; a < 3; ++a) { root_lo[a] = (lo[a] < root_lo[a]) ? lo[a] : root_lo[a]; root_hi[a] = (root_hi[a] < hi[a]) ? hi[a] : root_hi[a]; }
    }
    std::string err;
    hipError_t e = rtd::refit_bvh2(const_cast<DBvhNode*>(r->dev.bvh), (uint32_t)r->n_bvh, roots.data(), pads.data(), (uint32_t)roots.size(), r->dev.tris, g.stream, err);
    if (e != hipSuccess) return fail(RT_E_HIP, "rt_scene_refit: " + err);
    if (r->n_bvh) {
        std::vector<int32_t> roots4(roots.size(), -1);
        DBvh4Node* same = nullptr;
        uint32_t n4 = 0;
        e = rtd::collapse_to_bvh4(r->dev.bvh, (uint32_t)r->n_bvh, roots.data(), (uint32_t)roots.size(), g.stream, &same, &n4, roots4.data(), err,
                                  const_cast<DBvh4Node*>(r->dev.bvh4), (uint32_t)r->n_bvh4, r->bvh4_place);
        if (e != hipSuccess) return fail(RT_E_HIP, "rt_scene_refit: " + err);
        HIP_TRY(rtd::quantise_bvh4(r->dev.bvh4, (uint32_t)r->n_bvh4, const_cast<DBvh4Q*>(r->dev.bvh4q), g.stream));
    }
    HIP_TRY(hipMemcpyAsync(const_cast<DMesh*>(r->dev.meshes), r->host.meshes.data(), n_mesh * sizeof(DMesh), hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipEventRecord(g.ev1, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    if (device_ms) HIP_TRY(hipEventElapsedTime(device_ms, g.ev0, g.ev1));
    for (int a = 0; a < 3; ++a) { r->host.box_lo[a] = r->dev.box_lo[a] = root_lo[a]; r->host.box_hi[a] = r->dev.box_hi[a] = root_hi[a]; }
    { /* the margins of the light / square table were derived for the root box of the build: a box that outgrew them switches the table off */
        float far = 0.f;
        for (int a = 0; a < 3; ++a) far = std::max(far, std::max(std::fabs(root_lo[a]), std::fabs(root_hi[a])));
        if (!(std::sqrt(3.0f) * far + 0.002f <= r->dev.cert_radius)) r->dev.light_sq = nullptr;
    }
    g.order_last_slot = -1; /* a launch order or deep list probed before the vertices moved is not used again */
    guard.done = true;
    return RT_OK;
}

int rt_scene_destroy(rt_scene_t scene) {
    auto it = g.scenes.find(scene);
    if (it == g.scenes.end()) return fail(RT_E_HANDLE, "rt_scene_destroy: unknown scene");
    (void)hipSetDevice(g.device);
    (void)hipStreamSynchronize(g.stream);
    release(it->second);
    g.scenes.erase(it);
    return RT_OK;
}

int rt_scene_kdtree(rt_scene_t scene, int32_t* nodes, uint32_t* n_nodes, int32_t* refs, uint32_t* n_refs, float root_box[6]) {
    SceneRes* r = find(scene);
    if (!r) return fail(RT_E_HANDLE, "rt_scene_kdtree: unknown scene");
    const rtb::Built& b = r->host;
    if (nodes)
        for (size_t i = 0; i < b.kd.size(); ++i) {
            const DKdNode& k = b.kd[i];
            int32_t* o = nodes + 8 * i;
            o[0] = k.leaf; o[1] = k.axis; std::memcpy(&o[2], &k.cut, 4); o[3] = k.left; o[4] = k.right;
            o[5] = k.leaf ? k.first_ref : 0; o[6] = k.leaf ? k.n_refs : 0; o[7] = k.depth;
        }
    if (refs) {
        /* refs of the leaves in node order: leaf slices are already contiguous and ordered */
        size_t w = 0;
        for (const DKdNode& k : b.kd)
            if (k.leaf)
                for (int i = 0; i < k.n_refs; ++i) { refs[2 * w] = b.refs[k.first_ref + i].type; refs[2 * w + 1] = b.refs[k.first_ref + i].object; ++w; }
    }
    if (n_nodes) *n_nodes = (uint32_t)b.kd.size();
    if (n_refs) *n_refs = (uint32_t)b.refs.size();
    if (root_box) for (int a = 0; a < 3; ++a) { root_box[a] = b.box_lo[a]; root_box[3 + a] = b.box_hi[a]; }
    return RT_OK;
}

int rt_intersect_device(rt_scene_t scene, const float* d_org, const float* d_dir, size_t n, rt_hit* d_out, void* hip_stream) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_intersect: call rt_init first");
    SceneRes* r = find(scene);
    if (!r) return fail(RT_E_HANDLE, "rt_intersect: unknown scene");
    if (n == 0) return RT_OK;
    if (!d_org || !d_dir || !d_out) return fail(RT_E_INVALID, "rt_intersect: null buffer");
    if (n > 0x7fffffffull * 256) return fail(RT_E_LIMIT, "rt_intersect: too many rays for one launch");
    hipStream_t st = (hipStream_t)hip_stream; /* NULL = HIP's default stream */
    RT_DISPATCH(r->mode, k_intersect, dim3(blocks_for(n)), st, r->dev, d_org, d_dir, n, d_out);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_intersect(rt_scene_t scene, const float* org, const float* dir, size_t n, rt_hit* out) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_intersect: call rt_init first");
    if (n == 0) return RT_OK;
    if (!org || !dir || !out) return fail(RT_E_INVALID, "rt_intersect: null buffer");
    HIP_TRY(hipSetDevice(g.device));
    Scratch tmp;
    void *d_o, *d_d, *d_h;
    int rc;
    if ((rc = tmp.get(&d_o, 12 * n)) || (rc = tmp.get(&d_d, 12 * n)) || (rc = tmp.get(&d_h, sizeof(rt_hit) * n))) return rc;
    HIP_TRY(hipMemcpyAsync(d_o, org, 12 * n, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(d_d, dir, 12 * n, hipMemcpyHostToDevice, g.stream));
    if ((rc = rt_intersect_device(scene, (const float*)d_o, (const float*)d_d, n, (rt_hit*)d_h, g.stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_h, sizeof(rt_hit) * n, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return RT_OK;
}

int rt_mesh_intersect(rt_scene_t scene, uint32_t mesh, const float* org, const float* dir, size_t n, rt_hit* out) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_mesh_intersect: call rt_init first");
    SceneRes* r = find(scene);
    if (!r) return fail(RT_E_HANDLE, "rt_mesh_intersect: unknown scene");
    if (mesh >= r->dev.n_meshes) return fail(RT_E_INVALID, "rt_mesh_intersect: no such mesh");
    if (n == 0) return RT_OK;
    if (!org || !dir || !out) return fail(RT_E_INVALID, "rt_mesh_intersect: null buffer");
    HIP_TRY(hipSetDevice(g.device));
    Scratch tmp;
    void *d_o, *d_d, *d_h;
    int rc;
    if ((rc = tmp.get(&d_o, 12 * n)) || (rc = tmp.get(&d_d, 12 * n)) || (rc = tmp.get(&d_h, sizeof(rt_hit) * n))) return rc;
    HIP_TRY(hipMemcpyAsync(d_o, org, 12 * n, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(d_d, dir, 12 * n, hipMemcpyHostToDevice, g.stream));
    hipLaunchKernelGGL(k_mesh_intersect, dim3(blocks_for(n)), dim3(256), 0, g.stream, r->dev, mesh, (const float*)d_o,
                       (const float*)d_d, n, (rt_hit*)d_h);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_h, sizeof(rt_hit) * n, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return RT_OK;
}

int rt_primitive_intersect(rt_scene_t scene, int type, uint32_t object, const float* org, const float* dir, size_t n, rt_hit* out) {
    if (type == RT_MESH) return rt_mesh_intersect(scene, object, org, dir, n, out);
    if (!g.ready) return fail(RT_E_NOINIT, "rt_primitive_intersect: call rt_init first");
    SceneRes* r = find(scene);
    if (!r) return fail(RT_E_HANDLE, "rt_primitive_intersect: unknown scene");
    if (type != RT_SPHERE && type != RT_SQUARE) return fail(RT_E_INVALID, "rt_primitive_intersect: type must be RT_MESH, RT_SPHERE or RT_SQUARE");
    if (object >= (type == RT_SPHERE ? r->dev.n_spheres : r->dev.n_squares)) return fail(RT_E_INVALID, "rt_primitive_intersect: no such object");
    if (n == 0) return RT_OK;
    if (!org || !dir || !out) return fail(RT_E_INVALID, "rt_primitive_intersect: null buffer");
    HIP_TRY(hipSetDevice(g.device));
    Scratch tmp;
    void *d_o, *d_d, *d_h;
    int rc;
    if ((rc = tmp.get(&d_o, 12 * n)) || (rc = tmp.get(&d_d, 12 * n)) || (rc = tmp.get(&d_h, sizeof(rt_hit) * n))) return rc;
    HIP_TRY(hipMemcpyAsync(d_o, org, 12 * n, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(d_d, dir, 12 * n, hipMemcpyHostToDevice, g.stream));
    hipLaunchKernelGGL(k_prim_intersect, dim3(blocks_for(n)), dim3(256), 0, g.stream, r->dev, type, object, (const float*)d_o, (const float*)d_d, n,
                       (rt_hit*)d_h);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_h, sizeof(rt_hit) * n, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return RT_OK;
}

int rt_generate_rays(const rt_camera* cam, const float* u, const float* v, size_t n, float* org, float* dir) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_generate_rays: call rt_init first");
    if (!cam) return fail(RT_E_INVALID, "rt_generate_rays: null camera");
    if (n == 0) return RT_OK;
    if (!u || !v || !org || !dir) return fail(RT_E_INVALID, "rt_generate_rays: null buffer");
    DCamera dc;
    std::string err;
    if (!rtb::invert_camera(*cam, dc, err)) return fail(RT_E_INVALID, err);
    HIP_TRY(hipSetDevice(g.device));
    Scratch tmp;
    void *d_u, *d_v, *d_o, *d_d;
    int rc;
    if ((rc = tmp.get(&d_u, 4 * n)) || (rc = tmp.get(&d_v, 4 * n)) || (rc = tmp.get(&d_o, 12 * n)) || (rc = tmp.get(&d_d, 12 * n))) return rc;
    HIP_TRY(hipMemcpyAsync(d_u, u, 4 * n, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(d_v, v, 4 * n, hipMemcpyHostToDevice, g.stream));
    hipLaunchKernelGGL(k_generate, dim3(blocks_for(n)), dim3(256), 0, g.stream, dc, (const float*)d_u, (const float*)d_v, n,
                       (float*)d_o, (float*)d_d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(org, d_o, 12 * n, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipMemcpyAsync(dir, d_d, 12 * n, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return RT_OK;
}

int rt_trace(rt_scene_t scene, const float* org, const float* dir, size_t n, const uint32_t* pixel, const uint32_t* sample,
             uint32_t seed, uint32_t k0, float* rgb, rt_stats* stats) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_trace: call rt_init first");
    SceneRes* r = find(scene);
    if (!r) return fail(RT_E_HANDLE, "rt_trace: unknown scene");
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (n == 0) return RT_OK;
    if (!org || !dir || !pixel || !sample || !rgb) return fail(RT_E_INVALID, "rt_trace: null buffer");
    HIP_TRY(hipSetDevice(g.device));
    Scratch tmp;
    void *d_o, *d_d, *d_p, *d_s, *d_c, *d_cnt;
    int rc;
    if ((rc = tmp.get(&d_o, 12 * n)) || (rc = tmp.get(&d_d, 12 * n)) || (rc = tmp.get(&d_p, 4 * n)) || (rc = tmp.get(&d_s, 4 * n)) ||
        (rc = tmp.get(&d_c, 12 * n)) || (rc = tmp.get(&d_cnt, RT_COUNTER_WORDS * 8)))
        return rc;
    HIP_TRY(hipMemcpyAsync(d_o, org, 12 * n, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(d_d, dir, 12 * n, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(d_p, pixel, 4 * n, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(d_s, sample, 4 * n, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemsetAsync(d_cnt, 0, RT_COUNTER_WORDS * 8, g.stream));
    HIP_TRY(hipEventRecord(g.ev0, g.stream));
    RT_DISPATCH(r->mode, k_trace, dim3(blocks_for(n)), g.stream, r->dev, (const float*)d_o, (const float*)d_d, n, (const uint32_t*)d_p,
                (const uint32_t*)d_s, seed, k0, (float*)d_c, (uint64_t*)d_cnt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(g.ev1, g.stream));
    std::vector<uint64_t> slots(RT_COUNTER_WORDS);
    uint64_t cnt[RT_COUNTER_FIELDS];
    HIP_TRY(hipMemcpyAsync(rgb, d_c, 12 * n, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipMemcpyAsync(slots.data(), d_cnt, RT_COUNTER_WORDS * 8, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    rt_counters_total(slots.data(), cnt);
    if (stats) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g.ev0, g.ev1));
        stats->rays = cnt[0]; stats->samples = cnt[1]; stats->nodes = cnt[2]; stats->kernel_ms = ms;
        stats->bvh_nodes = cnt[3]; stats->bvh_tris = cnt[4]; stats->rays_skipped = cnt[5];
    }
    return RT_OK;
}

void rt_normalize_dirs(const float* dir, size_t n, float* out) {
    for (size_t i = 0; i < n; ++i) {
        float x = dir[3 * i], y = dir[3 * i + 1], z = dir[3 * i + 2];
        float L = std::sqrt(x * x + y * y + z * z);
        out[3 * i] = x / L; out[3 * i + 1] = y / L; out[3 * i + 2] = z / L;
    }
}

void rt_counters_total(const uint64_t* h, uint64_t out[RT_COUNTER_FIELDS]) {
    for (int k = 0; k < RT_COUNTER_FIELDS; ++k) out[k] = 0;
    for (size_t s = 0; s < RT_COUNTER_SLOTS; ++s)
        for (int k = 0; k < RT_COUNTER_FIELDS; ++k) out[k] += h[s * RT_COUNTER_STRIDE + k];
}

size_t rt_tiles_per_rank(uint32_t w, uint32_t h, uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world) {
    if (!tile_w || !tile_h || !world || rank >= world) return 0;
    uint64_t tiles = (uint64_t)((w + tile_w - 1) / tile_w) * ((h + tile_h - 1) / tile_h);
    return (size_t)((tiles + world - 1 - rank) / world);
}

int rt_render_device(rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h, uint32_t spp, uint32_t seed, uint32_t tile_w,
                     uint32_t tile_h, uint32_t rank, uint32_t world, float* d_rgb_tiles, uint64_t* d_counters, void* hip_stream) {
    return rt_render_pass_device(scene, cam, w, h, spp, 0, spp, seed, tile_w, tile_h, rank, world, d_rgb_tiles, d_counters, hip_stream);
}

/* One pass of a render, enqueued on `st`.  static_mem = false: the camera goes to the next slot of the camera ring and the
 * launch order to the next workspace of the order ring (library allocations).  static_mem = true: both live in the static
 * `scratch` array of this device and nothing is allocated, on the device or on the host heap. */
static int render_pass(bool static_mem, rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h, uint32_t spp_pass,
                       uint32_t sample0, uint32_t spp_total, uint32_t seed, uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world,
                       float* d_rgb_tiles, uint64_t* d_counters, void* hip_stream) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_render: call rt_init first");
    SceneRes* r = find(scene);
    if (!r) return fail(RT_E_HANDLE, "rt_render: unknown scene");
    if (!cam || !d_rgb_tiles) return fail(RT_E_INVALID, "rt_render: null argument");
    if ((uint64_t)sample0 + spp_pass > spp_total) return fail(RT_E_INVALID, "rt_render: pass exceeds the total sample count");
    const uint32_t spp = spp_pass;
    RenderParams rp;
    int rc = make_render_params(w, h, spp, seed, tile_w, tile_h, rank, world, rp);
    if (rc) return rc;
    rp.sample0 = sample0; rp.spp_total = spp_total;
    if (static_mem && rp.n_local_tiles > RT_NOALLOC_MAX_TILES)
        return fail(RT_E_LIMIT, "rt_render_device_noalloc: more than RT_NOALLOC_MAX_TILES (131072) tiles in this rank's share; use larger tiles or rt_render_device");
    DCamera dc;
    {
        std::string err;
        if (!rtb::invert_camera(*cam, dc, err)) return fail(RT_E_INVALID, err);
    }
    if (rp.n_local_tiles == 0) return RT_OK;
    hipStream_t st = (hipStream_t)hip_stream; /* NULL = HIP's default stream */
    DCamera* d_cam;
    int cam_slot = -1, order_slot = -1;
    if (static_mem) {
        d_cam = (DCamera*)g.scratch_ptr;
        hipLaunchKernelGGL(k_put_camera, dim3(1), dim3(64), 0, st, dc, d_cam);
        HIP_TRY(hipGetLastError());
    } else if (g.cam_last_slot >= 0 && g.cam_last_stream == hip_stream && std::memcmp(&g.cam_last, cam, sizeof(rt_camera)) == 0) {
        cam_slot = g.cam_last_slot; /* uploaded earlier on this very stream: ordered before anything enqueued now */
        d_cam = g.cam_ring + cam_slot;
    } else {
        /* a new camera goes to the next ring slot; a launch of another stream that may still read that slot is waited for.
         * The copy source is pageable, so it is staged before the call returns. */
        cam_slot = (int)(g.cam_next++ % kCamSlots);
        d_cam = g.cam_ring + cam_slot;
        if (g.cam_used[cam_slot] && g.cam_stream[cam_slot] != hip_stream) HIP_TRY(hipStreamWaitEvent(st, g.cam_done[cam_slot], 0));
        g.cam_last_slot = -1; /* until the copy is known to be enqueued */
        HIP_TRY(hipMemcpyAsync(d_cam, &dc, sizeof(DCamera), hipMemcpyHostToDevice, st));
        g.cam_last = *cam; g.cam_last_slot = cam_slot; g.cam_last_stream = hip_stream;
    }
    const uint32_t* d_order = nullptr;
    DeepList dl;
    std::memset(&dl, 0, sizeof(dl));
    /* The probe (k_tile_class) serves two things.  Launch order: without bounces every tree has one node, nothing to order;
     * with a triangle BVH the order gains nothing (there the long waves are the ones whose shadow rays skim the mesh, which
     * the probe does not see).  Deep list: the pixels whose centre ray has a tree of wide_nodes nodes or more go to
     * k_render_wide; that needs memory for their samples' colours, so not in the allocation-free call. */
    const bool want_order = g.order_tiles && r->has_bounce && !(r->mode & RT_M_BVH) && rp.n_local_tiles > 1;
    /* Breadth-first workers pay where a node of a path is slow to reach -- a chain of dependent BVH fetches, or the per-ray
     * kd traversal -- and the launch is too small to hide the waves that walk hundreds of such nodes: measured on one eighth
     * of the 10 M-triangle frame, 16 spp per pass: 9 - 12 ms -> 7.5 ms per pass; on the many-spheres kd scene 49 -> 42 ms;
     * but 15 -> 21 ms when the same share renders all 64 spp in one launch, and 22 -> 28 ms on the bare Cornell box, whose
     * nodes cost a depth-first lane next to nothing.  wide = 1 applies that rule, wide = 2 uses the workers wherever
     * something bounces. */
    const uint64_t first_launch_waves = (uint64_t)rp.n_local_tiles * (((uint64_t)tile_w * tile_h + (spp <= 32 ? 64 / spp : 1) - 1) / (spp <= 32 ? 64 / spp : 1));
    const bool wide_pays = (r->mode & RT_M_KD) || ((r->mode & RT_M_BVH) && first_launch_waves <= 40ull * 4096ull);
    const bool want_wide = r->has_bounce && !static_mem && (g.wide == 2 || (g.wide == 1 && wide_pays));
    if (want_order || want_wide) {
        uint32_t* ws = nullptr;
        bool fresh = true;
        if (static_mem) ws = (uint32_t*)(g.scratch_ptr + RT_NOALLOC_CAM_BYTES);
        else {
            const uint64_t npx = (uint64_t)rp.n_local_tiles * tile_w * tile_h;
            Lib::OrderKey key;
            std::memset(&key, 0, sizeof(key));
            key.scene = scene; key.stream = hip_stream; key.cam = *cam; key.w = w; key.h = h; key.tile_w = tile_w; key.tile_h = tile_h; key.rank = rank; key.world = world;
            if (sample0 > 0 && g.order_last_slot >= 0 && std::memcmp(&key, &g.order_key, sizeof(key)) == 0 &&
                (!want_wide || (uint64_t)g.order_deep_cap[g.order_last_slot] * spp * 12 <= g.order_colors_bytes[g.order_last_slot])) {
                order_slot = g.order_last_slot; /* same stream as the probe that filled it: the later passes of a progressive render */
                fresh = false;
            } else {
                const unsigned slot = g.order_next++ % kOrderSlots;
                /* at most a quarter of the pixels (beyond that the launch is full of long waves anyway) and 512 MiB of colours */
                uint64_t cap = want_wide ? std::max<uint64_t>(64, npx / 4) : 0;
                if (want_wide) cap = std::max<uint64_t>(1, std::min<uint64_t>(cap, (512ull << 20) / ((uint64_t)spp * 12)));
                const size_t off_head = (2 * (size_t)rp.n_local_tiles * 4 + 15) & ~(size_t)15;
                const size_t off_pixels = off_head + 16, off_flags = off_pixels + (size_t)cap * 4;
                const size_t off_colors = (off_flags + (want_wide ? (size_t)npx : 0) + 15) & ~(size_t)15;
                const size_t colors_bytes = (size_t)cap * spp * 12;
                /* worker waves: at most what the chip holds at 4 waves per SIMD (the deep samples are most of the work of a launch
                 * that uses them; a worker without an item ends at once), and no more than there can be items */
                const uint32_t workers = (uint32_t)std::min<uint64_t>(RT_TREE_WORKERS, (cap * spp + RT_TREE_BATCH / 2 - 1) / (RT_TREE_BATCH / 2));
                const size_t off_tree = (off_colors + colors_bytes + 255) & ~(size_t)255;
                const size_t bytes = off_tree + (size_t)workers * RT_TREE_NODES * RT_TREE_WORDS * 4;
                if (g.order_used[slot] && g.order_stream[slot] != hip_stream) HIP_TRY(hipStreamWaitEvent(st, g.order_done[slot], 0));
                if (g.order_cap[slot] < bytes) {
                    /* hipFree waits for the device, so no launch still reads the old block */
                    if (g.order_ws[slot]) { HIP_TRY(hipFree(g.order_ws[slot])); g.order_ws[slot] = nullptr; g.order_cap[slot] = 0; }
                    HIP_TRY(hipMalloc((void**)&g.order_ws[slot], bytes));
                    g.order_cap[slot] = bytes;
                }
                g.order_off[slot][0] = off_head; g.order_off[slot][1] = off_pixels; g.order_off[slot][2] = off_flags; g.order_off[slot][3] = off_colors; g.order_off[slot][4] = off_tree;
                g.order_workers[slot] = workers;
                g.order_deep_cap[slot] = (uint32_t)cap; g.order_colors_bytes[slot] = colors_bytes;
                g.order_key = key;
                g.order_last_slot = order_slot = (int)slot;
            }
            unsigned char* mem = (unsigned char*)g.order_ws[order_slot];
            ws = (uint32_t*)mem;
            if (want_wide && g.order_deep_cap[order_slot] > 0) {
                dl.head = (uint32_t*)(mem + g.order_off[order_slot][0]);
                dl.pixels = (uint32_t*)(mem + g.order_off[order_slot][1]);
                dl.flags = mem + g.order_off[order_slot][2];
                dl.colors = (float*)(mem + g.order_off[order_slot][3]);
                dl.tree = (uint32_t*)(mem + g.order_off[order_slot][4]);
                dl.n_workers = g.order_workers[order_slot];
                dl.cap = g.order_deep_cap[order_slot];
                dl.min_nodes = g.wide_nodes;
                /* a new list starts empty; a list used again only restarts k_render_wide's queue */
                if (fresh) HIP_TRY(hipMemsetAsync(dl.head, 0, 16, st));
                else HIP_TRY(hipMemsetAsync(dl.head + 1, 0, 4, st));
            }
        }
        if (fresh) {
            /* one thread per pixel of the tile: a tile of at most 64 pixels is probed by a single wave (with 256 threads three
             * of four waves of every workgroup would only wait at its barriers and hold wave slots: 0.20 -> 0.07 ms per frame) */
            if (tile_w * tile_h <= 64u) RT_DISPATCH_T(r->mode, k_tile_class, 64, dim3(rp.n_local_tiles), st, r->dev, (const DCamera*)d_cam, rp, ws, dl);
            else RT_DISPATCH_T(r->mode, k_tile_class, 256, dim3(rp.n_local_tiles), st, r->dev, (const DCamera*)d_cam, rp, ws, dl);
            if (want_order) hipLaunchKernelGGL(k_tile_place, dim3(1), dim3(1024), 0, st, ws, rp.n_local_tiles);
            HIP_TRY(hipGetLastError());
        }
        if (want_order) d_order = ws + rp.n_local_tiles;
    }
    /* The samples of the call are cut so that every launch fills its waves: whole 64-sample chunks go to the
     * wave-per-pixel kernel, what is left to the packed kernel in pieces of at most 32 (2 pixels per wave,
     * then 64 / rest pixels per wave).  50 spp = 32 + 18 keeps 94 % of the lanes busy where one wave per
     * pixel keeps 78 %.  The launches continue each other's sums through the tile buffer like passes do. */
    uint32_t left = spp, s0 = sample0;
    uint32_t n_workers = dl.flags ? dl.n_workers : 0u; /* in the first launch of the pass only: they take all its samples */
    while (left > 0) {
        const uint32_t n = left >= 64 ? (left / 64) * 64 : (left > 32 ? 32 : left);
        const bool pack = n <= 32;
        const uint32_t px_per_wave = pack ? 64u / n : 1u;
        const uint64_t waves = (uint64_t)rp.n_local_tiles * (((uint64_t)tile_w * tile_h + px_per_wave - 1) / px_per_wave);
        rp.spp = n; rp.sample0 = s0;
        dim3 grid((unsigned)((waves + RT_RENDER_WAVES - 1) / RT_RENDER_WAVES) + n_workers);
        RT_DISPATCH_RENDER(r->mode, pack, n_workers > 0, grid, st, r->dev, (const DCamera*)d_cam, rp, d_rgb_tiles, d_counters, d_order, dl, n_workers, spp, sample0);
        HIP_TRY(hipGetLastError());
        s0 += n; left -= n;
        n_workers = 0;
    }
    if (dl.flags) { /* the sums of the listed pixels, from the colours the workers left */
        rp.spp = spp; rp.sample0 = sample0;
        hipLaunchKernelGGL(k_resolve_deep, dim3(blocks_for(dl.cap)), dim3(256), 0, st, rp, dl, d_rgb_tiles);
        HIP_TRY(hipGetLastError());
    }
    /* the ring slots this call read are busy until here */
    if (cam_slot >= 0) { HIP_TRY(hipEventRecord(g.cam_done[cam_slot], st)); g.cam_used[cam_slot] = true; g.cam_stream[cam_slot] = hip_stream; }
    if (order_slot >= 0) { HIP_TRY(hipEventRecord(g.order_done[order_slot], st)); g.order_used[order_slot] = true; g.order_stream[order_slot] = hip_stream; }
    return RT_OK;
}

int rt_render_pass_device(rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h, uint32_t spp_pass, uint32_t sample0,
                          uint32_t spp_total, uint32_t seed, uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world,
                          float* d_rgb_tiles, uint64_t* d_counters, void* hip_stream) {
    return render_pass(false, scene, cam, w, h, spp_pass, sample0, spp_total, seed, tile_w, tile_h, rank, world, d_rgb_tiles,
                       d_counters, hip_stream);
}

int rt_render_device_noalloc(rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h, uint32_t spp, uint32_t seed, uint32_t tile_w,
                             uint32_t tile_h, uint32_t rank, uint32_t world, float* d_rgb_tiles, uint64_t* d_counters, void* hip_stream) {
    return render_pass(true, scene, cam, w, h, spp, 0, spp, seed, tile_w, tile_h, rank, world, d_rgb_tiles, d_counters,
                       hip_stream);
}

int rt_render(rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h, uint32_t spp, uint32_t seed, uint32_t tile_w,
              uint32_t tile_h, uint32_t rank, uint32_t world, float* rgb_tiles, rt_stats* stats) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_render: call rt_init first");
    if (!rgb_tiles) return fail(RT_E_INVALID, "rt_render: null output");
    if (stats) std::memset(stats, 0, sizeof(*stats));
    size_t n_tiles = rt_tiles_per_rank(w, h, tile_w, tile_h, rank, world);
    size_t bytes = n_tiles * tile_w * tile_h * 3 * sizeof(float);
    HIP_TRY(hipSetDevice(g.device));
    Scratch tmp;
    void *d_out, *d_cnt;
    int rc;
    if ((rc = tmp.get(&d_out, bytes)) || (rc = tmp.get(&d_cnt, RT_COUNTER_WORDS * 8))) return rc;
    HIP_TRY(hipMemsetAsync(d_cnt, 0, RT_COUNTER_WORDS * 8, g.stream));
    HIP_TRY(hipEventRecord(g.ev0, g.stream));
    if ((rc = rt_render_device(scene, cam, w, h, spp, seed, tile_w, tile_h, rank, world, (float*)d_out, (uint64_t*)d_cnt, g.stream))) return rc;
    HIP_TRY(hipEventRecord(g.ev1, g.stream));
    std::vector<uint64_t> slots(RT_COUNTER_WORDS);
    uint64_t cnt[RT_COUNTER_FIELDS];
    if (bytes) HIP_TRY(hipMemcpyAsync(rgb_tiles, d_out, bytes, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipMemcpyAsync(slots.data(), d_cnt, RT_COUNTER_WORDS * 8, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    rt_counters_total(slots.data(), cnt);
    if (stats) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g.ev0, g.ev1));
        stats->rays = cnt[0]; stats->samples = cnt[1]; stats->nodes = cnt[2]; stats->kernel_ms = ms;
        stats->bvh_nodes = cnt[3]; stats->bvh_tris = cnt[4]; stats->rays_skipped = cnt[5];
    }
    return RT_OK;
}

/* ---- multi-GPU: one process per GPU, RCCL point-to-point gather ------------------------------------------- */
int rt_comm_unique_id(char id[RT_COMM_ID_BYTES]) {
    if (!id) return fail(RT_E_INVALID, "rt_comm_unique_id: null buffer");
    std::string err;
    if (!rtc::unique_id(id, err)) return fail(RT_E_HIP, "rt_comm_unique_id: " + err);
    return RT_OK;
}

int rt_comm_init(const char id[RT_COMM_ID_BYTES], int rank, int world) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_comm_init: call rt_init first");
    if (!id) return fail(RT_E_INVALID, "rt_comm_init: null id");
    HIP_TRY(hipSetDevice(g.device));
    std::string err;
    if (!rtc::init(id, rank, world, err)) return fail(RT_E_INVALID, "rt_comm_init: " + err);
    return RT_OK;
}

int rt_comm_destroy(void) {
    if (g.ready) { (void)hipSetDevice(g.device); (void)hipStreamSynchronize(g.stream); }
    rtc::destroy();
    return RT_OK;
}

int rt_comm_rank(void) { return rtc::rank(); }
int rt_comm_world(void) { return rtc::world(); }

int rt_gather_tiles_device(const float* d_mine, size_t floats_per_rank, float* d_gathered, int root, void* hip_stream) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_gather_tiles_device: call rt_init first");
    if (!rtc::active()) return fail(RT_E_NOINIT, "rt_gather_tiles_device: call rt_comm_init first");
    if (!d_mine || (rtc::rank() == root && !d_gathered)) return fail(RT_E_INVALID, "rt_gather_tiles_device: null buffer");
    std::string err;
    if (!rtc::gather(d_mine, floats_per_rank, d_gathered, root, (hipStream_t)hip_stream, err)) return fail(RT_E_HIP, "rt_gather_tiles_device: " + err);
    return RT_OK;
}

/* grows one of the sharded-render buffers; only a growing call allocates (and then waits for the device) */
static int grow_device(float** p, size_t* have, size_t want) {
    if (*have >= want && *p) return RT_OK;
    if (*p) { HIP_TRY(hipFree(*p)); *p = nullptr; *have = 0; }
    HIP_TRY(hipMalloc((void**)p, (want ? want : 1) * sizeof(float)));
    *have = want;
    return RT_OK;
}

/* Render + gather + un-tile, all enqueued on `st`.  On rank 0 the row-major image ends up in d_image (w * h * 3 floats, a
 * device pointer of the caller, or the library's own when null); d_counters may be null.  A rank whose render cannot be
 * enqueued still takes part in the exchange (its share arrives as zeros) and reports the error afterwards, so that one
 * failing rank does not leave the others waiting in the collective. */
static int sharded_enqueue(rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h, uint32_t spp, uint32_t seed, uint32_t tile_w,
                           uint32_t tile_h, float* d_image, uint64_t* d_counters, hipStream_t st, bool timed) {
    const int rank = rtc::rank(), world = rtc::world(); /* 0 / 1 without a communicator */
    if (!w || !h || !spp || !tile_w || !tile_h) return fail(RT_E_INVALID, "rt_render_sharded: zero size");
    if ((uint64_t)w * h > 0x7fffffffull / 3) return fail(RT_E_LIMIT, "rt_render_sharded: image too large");
    /* every rank's buffer has the size of rank 0's, the largest: the gathered block is regular */
    const size_t per_rank_tiles = rt_tiles_per_rank(w, h, tile_w, tile_h, 0, (uint32_t)world);
    const size_t per_rank = per_rank_tiles * tile_w * tile_h * 3;
    if (per_rank == 0) return fail(RT_E_INVALID, "rt_render_sharded: zero size");
    Lib::Sharded& sh = g.sh;
    int rc;
    if ((rc = grow_device(&sh.mine, &sh.mine_floats, per_rank))) return rc;
    if (rank == 0 && world > 1 && (rc = grow_device(&sh.all, &sh.all_floats, per_rank * (size_t)world))) return rc;
    if (rank == 0 && !d_image) {
        if ((rc = grow_device(&sh.image, &sh.image_floats, (size_t)w * h * 3))) return rc;
        d_image = sh.image;
    }
    for (hipEvent_t& e : sh.ev) if (!e) HIP_TRY(hipEventCreate(&e));
    if (timed) HIP_TRY(hipEventRecord(sh.ev[0], st));
    /* the padding tiles of a rank with fewer tiles than rank 0 are never written by the render: keep them defined */
    const size_t own_floats = rt_tiles_per_rank(w, h, tile_w, tile_h, (uint32_t)rank, (uint32_t)world) * tile_w * tile_h * 3;
    if (own_floats < per_rank) HIP_TRY(hipMemsetAsync(sh.mine + own_floats, 0, (per_rank - own_floats) * 4, st));
    int rc_render = rt_render_device(scene, cam, w, h, spp, seed, tile_w, tile_h, (uint32_t)rank, (uint32_t)world, sh.mine, d_counters, st);
    std::string render_err;
    if (rc_render) { render_err = g.error; (void)hipMemsetAsync(sh.mine, 0, per_rank * 4, st); }
    if (timed) HIP_TRY(hipEventRecord(sh.ev[1], st));
    if (world > 1) {
        std::string err;
        if (!rtc::gather(sh.mine, per_rank, sh.all, 0, st, err)) return fail(RT_E_HIP, "rt_render_sharded: " + err);
    }
    if (timed) HIP_TRY(hipEventRecord(sh.ev[2], st));
    if (rank == 0) {
        const uint32_t tiles_x = (w + tile_w - 1) / tile_w;
        hipLaunchKernelGGL(k_tiles_to_image, dim3(blocks_for(3 * (size_t)w * h)), dim3(256), 0, st, world > 1 ? sh.all : sh.mine, w, h, tile_w, tile_h,
                           tiles_x, (uint32_t)world, (uint32_t)per_rank_tiles, d_image);
        HIP_TRY(hipGetLastError());
    }
    if (timed) HIP_TRY(hipEventRecord(sh.ev[3], st));
    if (rc_render) return fail(rc_render, render_err);
    return RT_OK;
}

int rt_render_sharded_device(rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h, uint32_t spp, uint32_t seed, uint32_t tile_w,
                             uint32_t tile_h, float* d_image, uint64_t* d_counters, void* hip_stream) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_render_sharded_device: call rt_init first");
    if (rtc::rank() == 0 && !d_image) return fail(RT_E_INVALID, "rt_render_sharded_device: rank 0 needs the image buffer");
    HIP_TRY(hipSetDevice(g.device));
    return sharded_enqueue(scene, cam, w, h, spp, seed, tile_w, tile_h, d_image, d_counters, (hipStream_t)hip_stream, false);
}

int rt_render_sharded(rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h, uint32_t spp, uint32_t seed, uint32_t tile_w, uint32_t tile_h,
                      float* rgb_image, rt_stats* stats) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_render_sharded: call rt_init first");
    const int rank = rtc::rank();
    if (rank == 0 && !rgb_image) return fail(RT_E_INVALID, "rt_render_sharded: rank 0 needs the image buffer");
    if (stats) std::memset(stats, 0, sizeof(*stats));
    HIP_TRY(hipSetDevice(g.device));
    Lib::Sharded& sh = g.sh;
    if (!sh.counters) HIP_TRY(hipMalloc((void**)&sh.counters, RT_COUNTER_WORDS * 8));
    HIP_TRY(hipMemsetAsync(sh.counters, 0, RT_COUNTER_WORDS * 8, g.stream));
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = sharded_enqueue(scene, cam, w, h, spp, seed, tile_w, tile_h, nullptr, sh.counters, g.stream, true);
    if (rc) { (void)hipStreamSynchronize(g.stream); return rc; }
    const size_t image_floats = (size_t)w * h * 3;
    double host_ms = 0.0;
    if (rank == 0) {
        /* device -> host: straight into the caller's buffer when it is pinned, else through the library's pinned block (a copy
         * into pageable memory is staged by the runtime in small pieces at a fraction of the link's rate) */
        hipPointerAttribute_t attr;
        const bool pinned = hipPointerGetAttributes(&attr, rgb_image) == hipSuccess && attr.type == hipMemoryTypeHost;
        (void)hipGetLastError(); /* an ordinary malloc'ed pointer makes the query fail: not an error of this call */
        if (pinned) HIP_TRY(hipMemcpyAsync(rgb_image, sh.image, image_floats * 4, hipMemcpyDeviceToHost, g.stream));
        else {
            if (sh.pinned_floats < image_floats) {
                if (sh.pinned) { HIP_TRY(hipHostFree(sh.pinned)); sh.pinned = nullptr; sh.pinned_floats = 0; }
                HIP_TRY(hipHostMalloc((void**)&sh.pinned, image_floats * 4, hipHostMallocDefault));
                sh.pinned_floats = image_floats;
            }
            HIP_TRY(hipMemcpyAsync(sh.pinned, sh.image, image_floats * 4, hipMemcpyDeviceToHost, g.stream));
        }
        HIP_TRY(hipEventRecord(sh.ev[4], g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        if (!pinned) {
            const auto h0 = std::chrono::steady_clock::now();
            std::memcpy(rgb_image, sh.pinned, image_floats * 4);
            host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
        }
    } else {
        HIP_TRY(hipEventRecord(sh.ev[4], g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
    }
    uint64_t slots[RT_COUNTER_WORDS];
    HIP_TRY(hipMemcpy(slots, sh.counters, RT_COUNTER_WORDS * 8, hipMemcpyDeviceToHost));
    float ms[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 4; ++k) HIP_TRY(hipEventElapsedTime(&ms[k], sh.ev[k], sh.ev[k + 1]));
    sh.last.render_ms = ms[0]; sh.last.gather_ms = ms[1]; sh.last.untile_ms = ms[2]; sh.last.copy_ms = ms[3]; sh.last.host_copy_ms = host_ms;
    sh.last.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) { /* this rank's share */
        uint64_t cnt[RT_COUNTER_FIELDS];
        rt_counters_total(slots, cnt);
        stats->rays = cnt[0]; stats->samples = cnt[1]; stats->nodes = cnt[2]; stats->kernel_ms = ms[0];
        stats->bvh_nodes = cnt[3]; stats->bvh_tris = cnt[4]; stats->rays_skipped = cnt[5];
    }
    return RT_OK;
}

int rt_sharded_last_times(rt_sharded_times* out) {
    if (!out) return fail(RT_E_INVALID, "rt_sharded_last_times: null output");
    *out = g.sh.last;
    return RT_OK;
}

int rt_device_count(int* n) {
    if (!n) return fail(RT_E_INVALID, "rt_device_count: null output");
    *n = 0;
    HIP_TRY(hipGetDeviceCount(n));
    return RT_OK;
}

int rt_tiles_to_rgb8_device(const float* d_gathered, uint32_t w, uint32_t h, uint32_t tile_w, uint32_t tile_h, uint32_t world,
                            uint8_t* d_rgb8, void* hip_stream) {
    if (!g.ready) return fail(RT_E_NOINIT, "rt_tiles_to_rgb8_device: call rt_init first");
    if (!d_gathered || !d_rgb8 || !w || !h || !tile_w || !tile_h || !world) return fail(RT_E_INVALID, "rt_tiles_to_rgb8_device: bad argument");
    if ((uint64_t)w * h > 0x7fffffffull) return fail(RT_E_LIMIT, "rt_tiles_to_rgb8_device: more than 2^31 pixels");
    const uint32_t tiles_x = (w + tile_w - 1) / tile_w;
    const uint32_t per_rank = (uint32_t)rt_tiles_per_rank(w, h, tile_w, tile_h, 0, world);
    hipLaunchKernelGGL(k_tiles_to_rgb8, dim3(blocks_for((size_t)w * h)), dim3(256), 0, (hipStream_t)hip_stream, d_gathered, w, h, tile_w, tile_h,
                       tiles_x, world, per_rank, d_rgb8);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_write_ppm(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h, int binary) {
    if (!path || !rgb8 || !w || !h) return fail(RT_E_INVALID, "rt_write_ppm: bad argument");
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(RT_E_INVALID, std::string("rt_write_ppm: could not open ") + path);
    const size_t n = (size_t)w * h;
    bool ok;
    if (binary) {
        ok = std::fprintf(f, "P6\n%u %u\n255\n", w, h) > 0 && std::fwrite(rgb8, 3, n, f) == n;
    } else {
        ok = std::fprintf(f, "P3\n%u %u\n255\n", w, h) > 0;
        std::string line;
        line.reserve(n * 12 + 2);
        for (size_t i = 0; i < n; ++i) {
            char buf[16];
            int k = std::snprintf(buf, sizeof buf, "%u %u %u ", (unsigned)rgb8[3 * i], (unsigned)rgb8[3 * i + 1], (unsigned)rgb8[3 * i + 2]);
            line.append(buf, (size_t)k);
        }
        line.push_back('\n');
        ok = ok && std::fwrite(line.data(), 1, line.size(), f) == line.size();
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RT_OK : fail(RT_E_INVALID, std::string("rt_write_ppm: write failed on ") + path);
}

int rt_tiles_to_image(const float* gathered, uint32_t w, uint32_t h, uint32_t tile_w, uint32_t tile_h, uint32_t world, float* rgb_image) {
    if (!gathered || !rgb_image || !w || !h || !tile_w || !tile_h || !world) return fail(RT_E_INVALID, "rt_tiles_to_image: bad argument");
    uint32_t tiles_x = (w + tile_w - 1) / tile_w, tiles_y = (h + tile_h - 1) / tile_h;
    size_t per_rank = rt_tiles_per_rank(w, h, tile_w, tile_h, 0, world); /* rank 0 holds the most */
    size_t tile_floats = (size_t)tile_w * tile_h * 3;
    for (uint32_t t = 0; t < tiles_x * tiles_y; ++t) {
        uint32_t rank = t % world, local = t / world;
        const float* src = gathered + ((size_t)rank * per_rank + local) * tile_floats;
        uint32_t x0 = rt_tile_column(t, tiles_x) * tile_w, y0 = (t / tiles_x) * tile_h;
        for (uint32_t yy = 0; yy < tile_h && y0 + yy < h; ++yy)
            for (uint32_t xx = 0; xx < tile_w && x0 + xx < w; ++xx)
                std::memcpy(rgb_image + 3 * ((size_t)(y0 + yy) * w + x0 + xx), src + 3 * ((size_t)yy * tile_w + xx), 12);
    }
    return RT_OK;
}

}  // extern "C"
