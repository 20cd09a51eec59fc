/* rt_bvh_device.h -- triangle BVH built on the GPU (see rt_bvh_device.hip). */
#ifndef RT_BVH_DEVICE_H
#define RT_BVH_DEVICE_H

#include <hip/hip_runtime.h>

#include <string>

#include "rt_types.h"

namespace rtd {

struct DeviceBvh {
    DBvhNode* nodes = nullptr; /* device array owned by the caller after a successful build (hipFree) */
    uint32_t n_nodes = 0;
    int32_t root = -1;         /* already offset by node_base */
    int max_depth = 0;         /* inner nodes on the longest root-to-leaf path */
    float sort_ms = 0.f, tree_ms = 0.f; /* device time: keys + sort + permutation, hierarchy + boxes */
};

/* Triangle records of one mesh, made on the device from the caller's vertex / index pools exactly as the host
 * scene builder makes them (vertices times 1.000001f, two edge differences, R/src/Mesh.h:220-234; global vertex
 * ids; tri_orig = identity).  d_tris / d_tri_orig point at the mesh's first record.  d_box: 8 words owned by
 * the caller and initialised with box_init(): the exact box of all record corners (v0, v0 + e1, v0 + e2) in an
 * order-preserving integer encoding (box_decode), then a flag that becomes 1 if a vertex index is
 * >= n_vertices (such a triangle gets an all-zero record).  Only enqueues work on the stream. */
void box_init(uint32_t box[8]);
void box_decode(const uint32_t box[8], float lo[3], float hi[3], bool& bad_index);
hipError_t make_mesh_tris(const float* d_positions, const uint32_t* d_triangles, uint32_t first_vertex, uint32_t n_vertices,
                          uint32_t n_triangles, DTri* d_tris, uint32_t* d_tri_orig, uint32_t* d_box, hipStream_t stream);

/* Builds the BVH of the triangles d_tris[first, first + count) -- records in the mesh's original order, as the
 * host scene builder lays them out -- entirely on the device: the records are permuted into leaf order in
 * place, d_tri_orig[first + i] receives the mesh-local original index of the record now at first + i, and the
 * nodes come back in a fresh device array whose child indices are offset by node_base (the position the array
 * will take in the scene's node table).  lo / hi: the mesh's padded box, pad: the padding every node box gets
 * (both as the host builder computes them).  Returns hipSuccess or the failing HIP status; err says where. */
hipError_t build_mesh_bvh(DTri* d_tris, uint32_t* d_tri_orig, uint32_t first, uint32_t count, const float lo[3],
                          const float hi[3], float pad, int32_t node_base, int max_levels, hipStream_t stream,
                          DeviceBvh& out, std::string& err);

/* The four-wide form of a table of binary trees (DBvh4Node, rt_types.h): every node at an even depth of its tree is
 * kept and takes over the children of its inner children.  d_bvh2[0, n2): the binary nodes of all meshes (child
 * indices into the same table); roots[0, n_roots): the root nodes, roots4 gets their positions in the new table.
 * The new table comes back in a fresh device array (hipFree by the caller) -- or, when `existing` is given (a table of
 * n_existing nodes made from the same trees: a refit), in that one.  d_place (device, one entry per wide node in the
 * order without it, or null) moves every node to the place it names; roots4 then holds the moved positions.
 * Blocks until done. */
hipError_t collapse_to_bvh4(const DBvhNode* d_bvh2, uint32_t n2, const int32_t* roots, uint32_t n_roots, hipStream_t stream,
                            DBvh4Node** d_bvh4, uint32_t* n4, int32_t* roots4, std::string& err, DBvh4Node* existing = nullptr,
                            uint32_t n_existing = 0, const uint32_t* d_place = nullptr);

/* Places for collapse_to_bvh4 that put the table into treelet order (depth first; the inner children of a node side by
 * side), computed on the host from the 64-byte table of a first collapse.  *d_place: fresh device array (hipFree by the caller). */
hipError_t treelet_places(const DBvh4Q* d_nodes, uint32_t n4, const int32_t* roots4, uint32_t n_roots, uint32_t** d_place, std::string& err);

/* The 64-byte nodes the rays fetch (DBvh4Q, rt_types.h) from the four-wide table, node for node; d_out holds n4 of them.
 * Enqueued on `stream`. */
hipError_t quantise_bvh4(const DBvh4Node* d_bvh4, uint32_t n4, DBvh4Q* d_out, hipStream_t stream);

/* Refit: the vertices moved, the triangles and the trees stay.
 * vertex_box: exact box of `count` raw positions into d_box (8 words, box_init / box_decode).
 * remake_mesh_tris: the `count` records at d_tris (any order -- after a build, leaf order) again from the vertex ids they
 *   carry and the new positions (d_positions = the scene's whole vertex pool of n_vertices), with the box of their corners
 *   in d_box as make_mesh_tris leaves it.
 * refit_bvh2: the child boxes of every node under the given roots, deepest level first, from the new records: a
 *   leaf's triangles padded by its tree's pad (pads[k] belongs to roots[k]), an inner child's two boxes joined.
 *   Blocks until done.  The four-wide table follows with collapse_to_bvh4(.., existing table). */
hipError_t vertex_box(const float* d_positions, uint32_t count, uint32_t* d_box, hipStream_t stream);
hipError_t remake_mesh_tris(const float* d_positions, uint32_t n_vertices, DTri* d_tris, uint32_t count, uint32_t* d_box, hipStream_t stream);
hipError_t refit_bvh2(DBvhNode* d_bvh2, uint32_t n2, const int32_t* roots, const float* pads, uint32_t n_roots, const DTri* d_tris, hipStream_t stream,
                      std::string& err);

}  // namespace rtd
#endif
