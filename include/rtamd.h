/*
 * rtamd.h -- C-ABI of the MI355X ray-intersection + shading hot path.
 *
 * The reference program (R/ = HAI719I_Raytracer/) has no FFI layer: main.cpp calls C++ member
 * functions of Scene / Camera / Mesh directly.  This header is the boundary a maintainer binds
 * at those call sites; every entry point names the reference interface it stands in for.
 *
 * Conventions: plain C types only; every function returns 0 on success and a negative
 * RT_E_* code on failure (rt_last_error() gives the text); no exceptions cross the boundary;
 * output buffers are caller-owned; calls are blocking unless the name ends in _device (then
 * the work is enqueued on the given HIP stream -- NULL is HIP's default stream -- and the pointers
 * are device pointers);
 * thread-compatible, not thread-safe; one process drives one GPU (rt_init picks it).
 */
#ifndef RTAMD_H
#define RTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------------------------ */
#define RT_OK 0
#define RT_E_INVALID (-1)   /* bad argument / descriptor                                   */
#define RT_E_NOINIT (-2)    /* rt_init not called                                          */
#define RT_E_HIP (-3)       /* a HIP runtime call failed (no GPU, out of memory, fault ..) */
#define RT_E_HANDLE (-4)    /* unknown scene handle                                        */
#define RT_E_LIMIT (-5)     /* size beyond a documented limit                              */

/* ---- object type codes: R/src/Scene.h:20-22 (MESH 0, SPHERE 1, SQUARE 2) ------------------ */
#define RT_MESH 0
#define RT_SPHERE 1
#define RT_SQUARE 2

/* ---- material type codes: R/src/Material.h:10-14 ----------------------------------------- */
#define RT_MAT_DIFFUSE 0
#define RT_MAT_GLASS 1
#define RT_MAT_MIRROR 2

/* ---- light type codes: R/src/Scene.h:26-29 ------------------------------------------------ */
#define RT_LIGHT_SPHERICAL 0
#define RT_LIGHT_QUAD 1

/* ---- RNG modes --------------------------------------------------------------------------- */
#define RT_RNG_COUNTER 0 /* seeded counter-based generator (rt_rand31 below)                */
#define RT_RNG_LIBC 1    /* glibc rand() from srand(1): CPU oracle only (sequential stream) */

/* Material: R/src/Material.h:17-34.  shininess is the reference's double narrowed to float the
 * way Scene::getMaterialProperties does (R/src/Scene.h:283-301). */
typedef struct rt_material {
    float diffuse[3];
    float specular[3];
    float shininess;
    float index_medium;
    int32_t type; /* RT_MAT_* */
} rt_material;

/* Sphere: R/src/Sphere.h:36-42 (m_center, m_radius) + its Mesh::material. */
typedef struct rt_sphere {
    float center[3];
    float radius;
    int32_t material; /* index into rt_scene_desc.materials */
} rt_sphere;

/* Square: the four vertices.position written by Square::setQuad and the Mesh transforms
 * (R/src/Square.h:30-61, R/src/Mesh.h:128-173).  intersect reads 0,1,3; the kd builder 0..2;
 * computeGlobalAABB all four. */
typedef struct rt_square {
    float v[4][3];
    int32_t material;
} rt_square;

/* Mesh: a slice of the shared vertex / triangle pools.  aabb_* is Mesh::aabb
 * (R/src/Mesh.h:108): (+inf, -inf) until Mesh::computeAABB ran, exactly as in the reference. */
typedef struct rt_mesh {
    uint32_t first_vertex, n_vertices;     /* into positions / normals (xyz triples)        */
    uint32_t first_triangle, n_triangles;  /* into triangles (index triples, mesh-local)    */
    int32_t material;
    float aabb_min[3], aabb_max[3];
} rt_mesh;

/* Light: R/src/Scene.h:32-46.  For a quad light only triangle (0,1,2) of Light::quad is ever
 * sampled (R/src/Scene.h:271-275), so p0..p2 are those three vertex positions. */
typedef struct rt_light {
    int32_t type; /* RT_LIGHT_* */
    float pos[3];
    float color[3]; /* Light::material */
    float p0[3], p1[3], p2[3];
} rt_light;

/* Flat scene: the state a reference Scene holds after a setup_* call. */
typedef struct rt_scene_desc {
    uint32_t n_spheres, n_squares, n_meshes, n_lights, n_materials;
    const rt_sphere* spheres;
    const rt_square* squares;
    const rt_mesh* meshes;
    const rt_light* lights;
    const rt_material* materials;
    uint32_t n_vertices, n_triangles;
    const float* positions;    /* 3 * n_vertices                                             */
    const float* normals;      /* 3 * n_vertices (Mesh::recomputeNormals output)             */
    const uint32_t* triangles; /* 3 * n_triangles, indices local to the owning mesh          */
    /* kd-tree parameters: R/src/Scene.h:96-100 (15, 30, 0.01f) */
    int32_t kd_max_depth;
    int32_t kd_max_leaf;
    float kd_epsilon;
} rt_scene_desc;

/* Camera state as the reference leaves it in OpenGL: the PROJECTION and MODELVIEW matrices
 * (column-major float, what glGetDoublev widens to double in R/src/matrixUtilities.h:188-217)
 * and GL_DEPTH_RANGE near. */
typedef struct rt_camera {
    float projection[16];
    float modelview[16];
    float depth_near;
} rt_camera;

/* One closest-hit record = what Scene::computeIntersection (R/src/Scene.h:169-205) and
 * Mesh::intersect (R/src/Mesh.h:206-281) return that is well defined. */
typedef struct rt_hit {
    int32_t hit;    /* intersectionExists                                                   */
    int32_t type;   /* RT_MESH / RT_SPHERE / RT_SQUARE, -1 on a miss                        */
    int32_t object; /* objectIndex, -1 on a miss                                            */
    int32_t tri;    /* tIndex of the closest triangle for a mesh hit, else -1               */
    float t;        /* FLT_MAX on a miss                                                    */
    float u, v;     /* mesh hit: barycentrics (w1 = u, w2 = v, w0 = 1 - u - v), else 0      */
    float pad0_;
    float p[3];     /* intersection point                                                   */
    float pad1_;
    float n[3];     /* unit normal                                                          */
    float pad2_;
} rt_hit;

typedef struct rt_stats {
    uint64_t rays;        /* closest-hit queries issued (primary + shadow + bounce)         */
    uint64_t samples;     /* primary samples traced                                         */
    uint64_t nodes;       /* shaded path nodes                                              */
    double kernel_ms;     /* device time of the render kernels (HIP events on the stream)   */
    uint64_t bvh_nodes;   /* triangle-BVH nodes fetched (64 B each)                          */
    uint64_t bvh_tris;    /* triangles tested inside BVH leaves (48 B each)                  */
    uint64_t rays_skipped; /* of `rays`: shadow queries the reference issues for a glass node whose Phong colour it then
                            * overwrites (R/src/Scene.h:626 vs :705) -- counted in `rays`, which stays the number of
                            * Scene::computeIntersection calls of the reference, but not traced: rays - rays_skipped
                            * queries were actually evaluated */
} rt_stats;

/* Work counters on the device: RT_COUNTER_SLOTS slots of RT_COUNTER_STRIDE uint64 each
 * ({rays, samples, nodes, bvh_nodes, bvh_tris, rays_skipped, 0...}); a workgroup adds into slot (workgroup id % RT_COUNTER_SLOTS), so the
 * atomics of a launch spread over many cache lines instead of serialising on one.  The totals are
 * the sums over the slots (rt_counters_total).  The caller zeroes the buffer. */
#define RT_COUNTER_SLOTS 1024
#define RT_COUNTER_STRIDE 8
#define RT_COUNTER_FIELDS 6
#define RT_COUNTER_WORDS (RT_COUNTER_SLOTS * RT_COUNTER_STRIDE)

typedef uint64_t rt_scene_t;

/* ---- shared counter-based RNG -------------------------------------------------------------
 * Replaces the reference's unseeded rand() (R/main.cpp:186-187, R/src/Scene.h:259-260).
 * A sample is identified by (seed, pixel = x + y*w, sample index s); the k-th rand() call the
 * reference would make while tracing that sample returns rt_rand31(rt_rng_key(..), k), a value
 * in [0, 2^31) like rand().  Integer-only, so CPU and GPU agree bit for bit. */
#if defined(__HIPCC__)
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif
RT_HD static inline uint32_t rt_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
RT_HD static inline uint32_t rt_rng_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    return rt_mix32(seed ^ rt_mix32(pixel ^ rt_mix32(sample + 0x9E3779B9U)));
}
RT_HD static inline uint32_t rt_rand31(uint32_t key, uint32_t k) {
    return rt_mix32(key + k * 0x9E3779B9U) >> 1;
}

/* ---- lifecycle ---------------------------------------------------------------------------- */
/* Select HIP device `device` for this process and create the library's stream.
 * No reference counterpart (the reference has no device). */
int rt_init(int device);
int rt_shutdown(void);
const char* rt_last_error(void);

/* Scheduling switches of the render calls; none of them changes a result (GPU tests compare the floats).  No
 * reference counterpart.  Each also has an environment variable read by rt_init.
 *   "tile_order" 0 / 1 (RTAMD_TILE_ORDER): launch the tiles deepest-path-first (default 1) or in index order
 *   "wide"       0 / 1 / 2 (RTAMD_WIDE): pixels with deep path trees are expanded breadth first by worker waves: never,
 *                where it pays (default: scenes with a kd-tree of internal nodes, and launches of a scene with a triangle
 *                BVH too small to hide a long wave), wherever something bounces
 *   "wide_nodes" 1 .. 24 (RTAMD_WIDE_NODES): a pixel counts as deep when the path tree of its centre ray, or of one of
 *                its corners, has at least this many nodes (default 12)
 *   "treelet_order" 0 / 1 (RTAMD_TREELET_ORDER): scenes created after the call keep the nodes of their triangle trees depth
 *                first with the inner children of a node side by side (default 0: measured without effect on frame times) */
int rt_set_option(const char* name, int64_t value);

/* Device self-test of the library's shortened divide / sqrt sequences: normalises `n` pseudo-random
 * vectors (components with random signs, mantissas and exponents in [2^-44, 2^44], i.e. across and
 * beyond the range the fast path accepts) both ways and counts results whose bits differ from the
 * IEEE operators `/` and sqrt.  *mismatches must come back 0.  No reference counterpart. */
int rt_selftest_arith(uint64_t n, uint32_t seed, uint64_t* mismatches);

/* ---- scene -------------------------------------------------------------------------------- */
/* New vertex positions and normals for the meshes of a scene whose triangles stay the same (a deforming or moving
 * mesh): the triangle records, the boxes of every BVH node (binary and four-wide tables) and the bounds of the meshes
 * and of the scene are recomputed on the device; the trees keep their shape and the records their order.
 * positions / normals: n_vertices x 3 floats laid out like rt_scene_desc.positions / .normals (host pointers),
 * n_vertices as at creation.  Queries after the call answer for the new geometry exactly as a scene created from it
 * would (same hit records, same images); a tree refitted over a strongly deformed mesh is merely slower to walk.
 * Only for scenes whose object kd-tree is one leaf because of their object count (at most kd_max_leaf objects, 30 in the
 * reference): RT_E_INVALID otherwise.  device_ms (may be null):
 * GPU time of the call.  No counterpart in the reference, where nothing moves: it is the companion of the tree build
 * on the GPU, for callers that animate a mesh between frames.  Blocks until done; not to be called while a render of
 * the scene is in flight. */
int rt_scene_refit(rt_scene_t scene, const float* positions, const float* normals, uint32_t n_vertices, float* device_ms);

/* Copy the scene, build the object-level kd-tree (R/src/KDTree.cpp:363-519, done on the host as
 * Scene::buildKDTree does) and the per-mesh triangle BVH, upload everything.  Stands in for the
 * state left by Scene::setup_* + Scene::buildKDTree (R/src/Scene.h:112-115, 722-1048). */
int rt_scene_create(const rt_scene_desc* desc, rt_scene_t* out);
int rt_scene_destroy(rt_scene_t scene);

/* rt_scene_create with options.  RT_BUILD_BVH_DEVICE: everything per triangle is done on the GPU from the
 * caller's vertex / index arrays -- the triangle records (R/src/Mesh.h:220-234 hoisted out of the ray loop),
 * the mesh bounds and the per-mesh BVHs (Morton order, top-down cuts; milliseconds for millions of triangles)
 * -- instead of by the host's loops and its binned-SAH builder (a somewhat better tree, seconds for 10 M
 * triangles).  Hit records do not depend on the choice: the reference's Mesh::intersect is a loop over all
 * triangles (R/src/Mesh.h:225), either tree only has to reach every triangle it would accept.
 * rt_scene_create(desc, out) is rt_scene_create_ex(desc, 0, out) for scenes below
 * RT_BUILD_AUTO_DEVICE_TRIANGLES triangles and rt_scene_create_ex(desc, RT_BUILD_BVH_DEVICE, out) from there
 * on; the environment variable RTAMD_BVH_BUILD=host|device overrides that choice. */
#define RT_BUILD_AUTO_DEVICE_TRIANGLES (1u << 20)
#define RT_MAX_TRIANGLES (1u << 28) /* triangles per scene: RT_E_LIMIT from there on */
#define RT_BUILD_BVH_DEVICE 1u
int rt_scene_create_ex(const rt_scene_desc* desc, uint32_t flags, rt_scene_t* out);

/* Host-only dry run of rt_scene_create (needs neither rt_init nor a GPU): validates the description, builds
 * the object kd-tree and the triangle BVHs, checks the BVHs (every triangle in exactly one leaf, boxes nested
 * and enclosing their triangles, depth within the device's traversal stack) and reports what was built. */
typedef struct rt_build_info {
    uint32_t kd_nodes, kd_max_depth;  /* object kd-tree (R/src/KDTree.cpp:363-519) */
    uint32_t bvh_meshes, bvh_nodes;   /* meshes that got a triangle BVH, total 64-byte nodes */
    uint32_t bvh_max_depth;           /* inner nodes on the longest root-to-leaf path (limit 32) */
    uint32_t kernel_variant;          /* bit 0: kd-tree has internal nodes, bit 1: some mesh has a BVH */
    uint64_t device_bytes;            /* size of the scene tables in HBM */
} rt_build_info;
int rt_scene_check(const rt_scene_desc* desc, rt_build_info* info);

/* The same checks on a scene that exists: copies the triangle and node tables back from the device and
 * verifies them (so it also covers BVHs built on the GPU).  info / times may be NULL.  times: what
 * rt_scene_create spent -- host_ms: host preparation (records, kd-tree, host BVH build if any);
 * device_sort_ms / device_tree_ms: device time of the GPU BVH build (keys + sort + permutation, hierarchy +
 * boxes; 0 for a host build); total_ms: the whole call including uploads. */
typedef struct rt_build_times {
    double host_ms, device_sort_ms, device_tree_ms, total_ms;
} rt_build_times;
int rt_scene_verify(rt_scene_t scene, rt_build_info* info, rt_build_times* times);

/* Flattened object kd-tree, for inspection / parity tests.  nodes: 8 int32 per node
 * {is_leaf, axis, cut(float bits), left, right, first_ref, n_refs, depth}; refs: 2 int32 per
 * reference {type, object}; root_box: min xyz, max xyz.  Pass NULL buffers to query counts. */
int rt_scene_kdtree(rt_scene_t scene, int32_t* nodes, uint32_t* n_nodes, int32_t* refs,
                    uint32_t* n_refs, float root_box[6]);

/* ---- batch closest hit: Scene::computeIntersection (R/src/Scene.h:169-205) over n rays ------
 * org/dir: 3*n floats = origin() and direction() of the reference's Ray objects: the Ray constructor has
 * already normalised the direction (R/src/Line.h:13-16) and the callee does not normalise again.
 * rt_normalize_dirs applies that constructor's normalisation to raw directions. */
int rt_intersect(rt_scene_t scene, const float* org, const float* dir, size_t n, rt_hit* out);
int rt_intersect_device(rt_scene_t scene, const float* d_org, const float* d_dir, size_t n,
                        rt_hit* d_out, void* hip_stream);

/* d[i] /= |d[i]| exactly as Line's constructor computes it (sqrt of the left-to-right sum of squares, three
 * true divides).  Host arithmetic, in place allowed (out == dir). */
void rt_normalize_dirs(const float* dir, size_t n, float* out);

/* ---- batch Mesh::intersect (R/src/Mesh.h:206-281) of mesh `mesh` alone: exposes tIndex ------ */
int rt_mesh_intersect(rt_scene_t scene, uint32_t mesh, const float* org, const float* dir,
                      size_t n, rt_hit* out);

/* ---- one object's own intersect routine: Sphere::intersect (R/src/Sphere.h:85-162), Square::intersect
 * (R/src/Square.h:63-115), Mesh::intersect (type RT_MESH = rt_mesh_intersect) -- without the scene's root box and
 * without the other objects.  hit / t / p / n as those routines leave them; Sphere's secondintersection, which nothing
 * in the reference reads, is not returned. */
int rt_primitive_intersect(rt_scene_t scene, int type, uint32_t object, const float* org, const float* dir,
                           size_t n, rt_hit* out);

/* ---- primary rays: screen_space_to_world_space_ray (R/src/matrixUtilities.h:218-222) --------
 * u, v: n floats each; org/dir: 3*n floats out (dir normalised once, as that function leaves it). */
int rt_generate_rays(const rt_camera* cam, const float* u, const float* v, size_t n, float* org,
                     float* dir);

/* ---- batch Scene::rayTrace (R/src/Scene.h:716-720) under the counter RNG -------------------
 * Ray i (origin / unit direction as for rt_intersect) is traced as sample (pixel[i], sample[i]); its rand()
 * calls start at index k0 (2 when the two jitter draws of R/main.cpp:186-187 precede it, 0 otherwise).
 * rgb: 3*n floats. */
int rt_trace(rt_scene_t scene, const float* org, const float* dir, size_t n, const uint32_t* pixel,
             const uint32_t* sample, uint32_t seed, uint32_t k0, float* rgb, rt_stats* stats);

/* ---- full render: ray_trace_from_camera (R/main.cpp:170-221) without the PPM write ---------
 * The image is cut into tile_w x tile_h pixel tiles in tiles_x = ceil(w / tile_w) columns.  Tile t lies
 * in tile-row t / tiles_x at column (t % tiles_x + t / tiles_x) % tiles_x (rt_tile_column): every row
 * is rotated one tile against the row above, so that the tiles of one rank form diagonals whatever
 * tiles_x and world are and no rank is tied to fixed image columns (with plain row-major numbering
 * and tiles_x a multiple of world a rank renders vertical stripes, and the ranks whose stripes cross the
 * glass sphere take 40 % longer than the others).  This call renders the
 * tiles t with t % world == rank (rank 0, world 1 = whole image).  Output is tile-major:
 * local tile j (global tile rank + j*world) occupies floats [j*tile_w*tile_h*3, ...), row-major
 * RGB inside the tile; pixels outside the image are 0.  rgb_tiles must hold
 * rt_tiles_per_rank(..)*tile_w*tile_h*3 floats. */
static inline RT_HD uint32_t rt_tile_column(uint32_t t, uint32_t tiles_x) {
    return (t % tiles_x + t / tiles_x) % tiles_x;
}
size_t rt_tiles_per_rank(uint32_t w, uint32_t h, uint32_t tile_w, uint32_t tile_h, uint32_t rank,
                         uint32_t world);
int rt_render(rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h, uint32_t spp,
              uint32_t seed, uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world,
              float* rgb_tiles, rt_stats* stats);
int rt_render_device(rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h, uint32_t spp,
                     uint32_t seed, uint32_t tile_w, uint32_t tile_h, uint32_t rank,
                     uint32_t world, float* d_rgb_tiles, uint64_t* d_counters, void* hip_stream);
/* One pass of a progressive render: samples [sample0, sample0 + spp_pass) of the spp_total the finished
 * image has.  d_rgb_tiles carries the running float sum of R/main.cpp:191 between passes (pass 0 writes
 * it, later passes continue it in sample order, the pass that reaches spp_total divides by spp_total as
 * R/main.cpp:193 does), so any split into passes yields the floats of rt_render_device(spp_total).
 * Between passes the buffer holds sums: divide by the samples so far to display it.
 * rt_render_device(.., spp, ..) = rt_render_pass_device(.., spp, 0, spp, ..).
 * Both calls only enqueue work on hip_stream.  The library keeps the per-call device state (inverted camera,
 * tile launch order) in rings of 64 and 8 slots; a call that gets a slot whose last reader was enqueued on a
 * different stream makes its own stream wait for that reader (an event per slot), so any number of calls may be
 * in flight on any number of streams.
 * Memory and synchronisation: a render slot's workspace is allocated by the first call that uses it and grown (hipFree +
 * hipMalloc, which wait for the device) by a call that needs more; calls of the same size afterwards only enqueue.  Without
 * the breadth-first workers a slot holds 8 bytes per tile.  With them (scenes with kd internal nodes or a triangle BVH and a
 * bouncing sphere, option "wide") it also holds a byte per pixel, 12 bytes x spp per listed pixel (at most a quarter of the
 * pixels and 512 MiB) and 112 KiB of path-tree memory per worker wave (at most 4096 workers, the wave slots of the chip at
 * four waves per SIMD: 448 MiB): up to about 0.9 GiB per slot, and a caller that changes the camera every frame cycles
 * through all eight slots.  Blocks are kept until rt_shutdown.  rt_render_device_noalloc never allocates. */
int rt_render_pass_device(rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h,
                          uint32_t spp_pass, uint32_t sample0, uint32_t spp_total, uint32_t seed,
                          uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world,
                          float* d_rgb_tiles, uint64_t* d_counters, void* hip_stream);
/* rt_render_device for callers that cannot allocate device memory while they run (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments,
 * same tile-major output and same floats as rt_render_device, but the call allocates nothing -- no hipMalloc /
 * hipFree, no host-heap allocation on the success path: the inverted camera (passed to the device as a kernel
 * argument) and the workspace of the tile launch order live in one static array in device memory of
 * 512 B + 8 B x RT_NOALLOC_MAX_TILES = 1 MiB + 512 B, which every GPU the library runs on holds until the process
 * ends.  A rank's share of more than RT_NOALLOC_MAX_TILES tiles is refused with RT_E_LIMIT (3840 x 2160 in 16 x 16
 * tiles is 32 400).  Every call on a GPU uses that GPU's one array: calls that overlap on one GPU (two streams, or
 * two threads) race on it; calls on one stream are ordered by the stream.  Nothing in the array has to start at
 * zero -- each call rewrites what it reads. */
#define RT_NOALLOC_MAX_TILES (1u << 17)
int rt_render_device_noalloc(rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h, uint32_t spp,
                             uint32_t seed, uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world,
                             float* d_rgb_tiles, uint64_t* d_counters, void* hip_stream);
/* d_counters: NULL, or a zeroed device buffer of RT_COUNTER_WORDS uint64 (see above).  Host helper:
 * totals {rays, samples, nodes, bvh_nodes, bvh_tris, rays_skipped} of a counter buffer copied back to the host. */
void rt_counters_total(const uint64_t* h_counters, uint64_t out[RT_COUNTER_FIELDS]);

/* Scatter `world` gathered tile buffers (rank-major, each padded to max_tiles_per_rank tiles)
 * into a row-major w*h*3 image.  Host helper for the gather step. */
int rt_tiles_to_image(const float* gathered, uint32_t w, uint32_t h, uint32_t tile_w,
                      uint32_t tile_h, uint32_t world, float* rgb_image);

/* ---- several GPUs: one process per GPU, tiles gathered with RCCL over xGMI ----------------------------------------
 * No reference counterpart (the reference is one thread on one CPU).  Each process calls rt_init(its GPU); one of them
 * makes an id (rt_comm_unique_id) and hands its RT_COMM_ID_BYTES bytes to the others by any means (a file, a pipe,
 * MPI); all call rt_comm_init(id, rank, world) together.  RCCL is opened at run time by the first of these calls.
 * rt_render_sharded: every rank renders the tiles t with t % world == rank (as rt_render(.., rank, world, ..)), the
 * tile buffers go to rank 0 in one group of point-to-point transfers (a receive per peer on rank 0, a send on the
 * others: seven distinct xGMI links into rank 0 on an 8-GPU node), and rank 0 gets the row-major w*h*3 float image;
 * rgb_image is ignored on the other ranks.  stats describe this rank's share.  Without a communicator it renders the
 * whole image (rank 0 of 1).  Collective: all ranks call it with the same arguments.
 * rt_gather_tiles_device is that exchange alone for callers that keep their own device buffers: floats_per_rank floats
 * of every rank arrive rank-major in d_gathered on `root` (its own part copied), enqueued on hip_stream. */
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(char id[RT_COMM_ID_BYTES]);
int rt_comm_init(const char id[RT_COMM_ID_BYTES], int rank, int world);
int rt_comm_destroy(void);
int rt_comm_rank(void);
int rt_comm_world(void);
int rt_gather_tiles_device(const float* d_mine, size_t floats_per_rank, float* d_gathered, int root, void* hip_stream);
int rt_render_sharded(rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h, uint32_t spp, uint32_t seed,
                      uint32_t tile_w, uint32_t tile_h, float* rgb_image, rt_stats* stats);
/* The frame loop behind it keeps its device buffers, a pinned host block and its events for the life of the library (a
 * frame of the same size allocates nothing), un-tiles on the device and copies only the final row-major image to the host
 * -- directly when rgb_image is pinned memory (hipHostMalloc / hipHostRegister), else through the pinned block.  A rank
 * whose render fails still takes part in the exchange (its share arrives as zeros) and returns its error afterwards, so
 * the other ranks are not left waiting.
 * rt_render_sharded_device: the same with everything enqueued on hip_stream and nothing copied to the host: on rank 0 the
 * row-major image is written to d_image (w * h * 3 floats of device memory; ignored elsewhere), d_counters may be null.
 * rt_sharded_last_times: device time of the phases of the latest rt_render_sharded call on this rank (HIP events), the
 * host-side copy out of the pinned block, and the wall time of the whole call.
 * rt_device_count: HIP devices visible to this process (a launcher checks it before its ranks join a communicator). */
typedef struct rt_sharded_times {
    double render_ms, gather_ms, untile_ms, copy_ms; /* device: render kernels, RCCL exchange, tile buffers -> image, image -> host */
    double host_copy_ms;                             /* pinned block -> rgb_image (0 when rgb_image is pinned itself) */
    double total_ms;
} rt_sharded_times;
int rt_render_sharded_device(rt_scene_t scene, const rt_camera* cam, uint32_t w, uint32_t h, uint32_t spp, uint32_t seed,
                             uint32_t tile_w, uint32_t tile_h, float* d_image, uint64_t* d_counters, void* hip_stream);
int rt_sharded_last_times(rt_sharded_times* out);
int rt_device_count(int* n);

/* ---- image output (R/main.cpp:204-216) ------------------------------------------------------
 * rt_tiles_to_rgb8_device: what rt_tiles_to_image does, on the device and fused with the reference's
 * quantisation (int)(255.f * min(1.f, c)) (R/main.cpp:213): gathered tile buffers (rank-major, each padded
 * to the tile count of rank 0) -> row-major 8-bit RGB, 3 * w * h bytes.  Enqueued on hip_stream.
 * rt_write_ppm: 8-bit RGB to a PPM file; binary = 0 writes the reference's ASCII P3 layout (header lines
 * "P3", "w h", "255", then "r g b " per pixel and a final newline), binary = 1 the same pixels as P6. */
int rt_tiles_to_rgb8_device(const float* d_gathered, uint32_t w, uint32_t h, uint32_t tile_w, uint32_t tile_h,
                            uint32_t world, uint8_t* d_rgb8, void* hip_stream);
int rt_write_ppm(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h, int binary);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_H */
