/* rt_build.h -- host-side scene preparation for the HIP path (see rt_build.cpp). */
#ifndef RT_BUILD_H
#define RT_BUILD_H

#include <string>
#include <vector>

#include "../../include/rtamd.h"
#include "rt_types.h"

namespace rtb {

struct Built {
    std::vector<DSphere> spheres;
    std::vector<DSquare> squares;
    std::vector<DMesh> meshes;
    std::vector<DTri> tris;
    std::vector<uint32_t> tri_orig;
    std::vector<DBvhNode> bvh;
    std::vector<float> normals;
    std::vector<DMaterial> materials;
    std::vector<DLight> lights;
    std::vector<DKdNode> kd;   /* kd[0] = root, pre-order */
    std::vector<DRef> refs;    /* leaf slices, list order of the reference */
    std::vector<DLeafRec> leaf_recs; /* with internal kd nodes: the records in reference order (see DeviceScene) */
    std::vector<DLightSq> light_sq; /* DeviceScene::light_sq */
    float cert_radius = 0.f;      /* DeviceScene::cert_radius */
    float box_lo[3], box_hi[3];
    float kd_epsilon = 0.01f;
};

/* meshes with more than kBvhMinTris triangles get a BVH; smaller ones keep the reference's loop */
const uint32_t kBvhMinTris = 64;
const int kBvhMaxDepth = 32; /* inner nodes on any root-to-leaf path = entries of the device's traversal stack */
/* defer_meshes: everything per triangle is left to the caller (the device path, rt_bvh_device.hip): meshes come
 * back with first_tri / n_tris / mat only, and tris, tri_orig, bvh and normals stay empty. */
bool build_scene(const rt_scene_desc& d, Built& out, std::string& err, bool defer_meshes = false);
/* Padded box and bounding sphere of a mesh from the exact box [lo, hi] of its triangle records' corners
 * (v0, v0 + e1, v0 + e2); returns the pad, which the node boxes of the mesh's BVH take as well. */
float mesh_bounds_from_box(const float lo[3], const float hi[3], uint32_t count, DMesh& mesh);
/* Walks every triangle BVH of a built scene: each triangle of a BVH mesh sits in exactly one leaf, leaves hold
 * 1..4 triangles, every child box contains the (padded) boxes below it and its triangles' vertices, and no
 * root-to-leaf path has more than kBvhMaxDepth inner nodes.  max_depth returns the longest path found. */
bool verify_bvh(const Built& b, int& max_depth, std::string& err);
/* The same walk over the four-wide trees the rays traverse (b.meshes[].bvh4_root into bvh4): every triangle of a BVH mesh in
 * exactly one leaf, boxes nested and enclosing their triangles; max_stack = the most entries a traversal can have pending. */
bool verify_bvh4(const Built& b, const std::vector<DBvh4Node>& bvh4, int& max_stack, std::string& err);
/* The quantised nodes against the full-precision ones they were made from: same children in the same slots, and every decoded
 * child box (fmaf(q, scale, lo), the traversal's expression) encloses the original one. */
bool verify_bvh4q(const std::vector<DBvh4Node>& bvh4, const std::vector<DBvh4Q>& quant, std::string& err);
bool invert_camera(const rt_camera& cam, DCamera& out, std::string& err);

}  // namespace rtb
#endif
