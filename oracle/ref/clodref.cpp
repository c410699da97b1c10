// clodref.cpp -- thin C ABI over the reference's own cluster-LOD builder.
// TEST INFRASTRUCTURE: loaded by tests/clodref_bridge.py only, which hands clodref_dag_build / clodref_dag_release to
// brmi_scene_create_with_dag_builder as the caller-supplied DAG builder.  Nothing under basicrenderer_amd/ loads it.
//
// What is compiled here is the reference's code where it lies under /root/reference (nothing is copied):
//   ThirdParty/meshoptimizer/src/*.cpp                        vendored meshoptimizer 1.0
//   BasicRenderer/include/ThirdParty/meshoptimizer/clusterlod.h   the reference's patched clusterlod (clodBuild)
// The driver below only (a) fills clodConfig the way the reference does (BR/src/Mesh/ClusterLODUtilities.cpp:5426-5458;
// BR/include/Mesh/ClusterLODTypes.h:189-192 for the defaults of the merge / partition settings) and (b) flattens the
// callback stream (one call per group, clodOutput) into arrays a C caller can read.
#include <cstdint>
#include <cstring>
#include <vector>

#include "brmi_scene.h"      // brmi_dag: the flat DAG arrays the scene generator consumes (this repository's header)
#include "meshoptimizer.h"
#define CLUSTERLOD_IMPLEMENTATION
#include "clusterlod.h"

namespace {
struct GroupOut { int32_t depth; float center[3], radius, error; uint32_t firstCluster, clusterCount; };
struct ClusterOut { int32_t group, refined; float center[3], radius, error; uint32_t vertexCount, triangleCount, firstVertex, firstTriangleByte; };
struct Result {
    std::vector<GroupOut> groups; std::vector<ClusterOut> clusters;
    std::vector<uint32_t> vertices;          // per cluster: original vertex index of every local vertex
    std::vector<uint8_t> triangles;          // per cluster: 3 local indices per triangle
};
}  // namespace

extern "C" {

struct clodref_result;   // opaque

// Builds the cluster-LOD DAG of one indexed triangle mesh (float3 positions, optional float3 normals as the simplification attribute).
clodref_result* clodref_build(const float* positions, size_t vertexCount, const uint32_t* indices, size_t indexCount, const float* normals) {
    Result* res = new Result();
    clodMesh mesh{};
    mesh.indices = indices; mesh.index_count = indexCount; mesh.vertex_count = vertexCount;
    mesh.vertex_positions = positions; mesh.vertex_positions_stride = 12;
    const float weights[3] = {0.5f, 0.5f, 0.5f};
    if (normals) { mesh.vertex_attributes = normals; mesh.vertex_attributes_stride = 12; mesh.attribute_weights = weights; mesh.attribute_count = 3; }
    // ClusterLODUtilities.cpp:5426-5458
    clodConfig config = clodDefaultConfig(128);
    config.max_vertices = 128; config.max_triangles = 128; config.min_triangles = 64;
    config.cluster_spatial = true; config.cluster_fill_weight = 0.5f; config.cluster_split_factor = 2.0f;
    config.partition_spatial = true; config.partition_sort = true;
    config.optimize_clusters = true; config.optimize_bounds = true;
    config.simplify_fallback_sloppy = true; config.simplify_error_factor_sloppy = 100.0f;
    config.simplify_fallback_permissive = false;
    config.simplify_error_merge_additive = 0.0f; config.simplify_error_merge_previous = 1.5f;
    config.partition_max_refined_groups = 8;
    config.partition_size = 384;             // max((512 * 3) / 4, partitionSizeFloor = 8)
    size_t splitCount = 0; config.partition_refined_split_count = &splitCount;

    clodBuild(config, mesh, [&](clodGroup group, const clodCluster* clusters, size_t count) -> int {
        GroupOut g{}; g.depth = group.depth;
        std::memcpy(g.center, group.simplified.center, 12); g.radius = group.simplified.radius; g.error = group.simplified.error;
        g.firstCluster = (uint32_t)res->clusters.size(); g.clusterCount = (uint32_t)count;
        const int id = (int)res->groups.size();
        for (size_t i = 0; i < count; i++) {
            const clodCluster& c = clusters[i];
            ClusterOut o{}; o.group = id; o.refined = c.refined;
            std::memcpy(o.center, c.bounds.center, 12); o.radius = c.bounds.radius; o.error = c.bounds.error;
            o.triangleCount = (uint32_t)(c.index_count / 3); o.firstVertex = (uint32_t)res->vertices.size(); o.firstTriangleByte = (uint32_t)res->triangles.size();
            std::vector<uint32_t> verts(c.vertex_count ? c.vertex_count : c.index_count);
            std::vector<uint8_t> tris(c.index_count);
            const size_t unique = clodLocalIndices(verts.data(), tris.data(), c.indices, c.index_count);
            o.vertexCount = (uint32_t)unique;
            res->vertices.insert(res->vertices.end(), verts.begin(), verts.begin() + unique);
            res->triangles.insert(res->triangles.end(), tris.begin(), tris.end());
            res->clusters.push_back(o);
        }
        res->groups.push_back(g);
        return id;
    });
    return reinterpret_cast<clodref_result*>(res);
}

void clodref_counts(const clodref_result* r, uint32_t* groups, uint32_t* clusters, uint32_t* vertexRefs, uint32_t* triangleBytes) {
    const Result* res = reinterpret_cast<const Result*>(r);
    *groups = (uint32_t)res->groups.size(); *clusters = (uint32_t)res->clusters.size(); *vertexRefs = (uint32_t)res->vertices.size(); *triangleBytes = (uint32_t)res->triangles.size();
}
// layouts: group = {i32 depth, f32 center[3], radius, error, u32 firstCluster, clusterCount} (32 B);
// cluster = {i32 group, refined, f32 center[3], radius, error, u32 vertexCount, triangleCount, firstVertex, firstTriangleByte} (44 B)
void clodref_copy(const clodref_result* r, void* groups, void* clusters, uint32_t* vertexRefs, uint8_t* triangles) {
    const Result* res = reinterpret_cast<const Result*>(r);
    std::memcpy(groups, res->groups.data(), res->groups.size() * sizeof(GroupOut));
    std::memcpy(clusters, res->clusters.data(), res->clusters.size() * sizeof(ClusterOut));
    std::memcpy(vertexRefs, res->vertices.data(), res->vertices.size() * 4);
    std::memcpy(triangles, res->triangles.data(), res->triangles.size());
}
void clodref_free(clodref_result* r) { delete reinterpret_cast<Result*>(r); }

// the same build with the brmi_dag_build_fn / brmi_dag_release_fn signatures (GroupOut / ClusterOut are brmi_dag_group / brmi_dag_cluster)
static_assert(sizeof(GroupOut) == sizeof(brmi_dag_group) && sizeof(ClusterOut) == sizeof(brmi_dag_cluster), "DAG record layouts");
int clodref_dag_build(void*, const float* positions, size_t vertexCount, const uint32_t* indices, size_t indexCount, const float* normals, brmi_dag* out) {
    Result* res = reinterpret_cast<Result*>(clodref_build(positions, vertexCount, indices, indexCount, normals));
    if (!res || !out) return -1;
    out->groups = reinterpret_cast<const brmi_dag_group*>(res->groups.data()); out->groupCount = (uint32_t)res->groups.size();
    out->clusters = reinterpret_cast<const brmi_dag_cluster*>(res->clusters.data()); out->clusterCount = (uint32_t)res->clusters.size();
    out->vertexRefs = res->vertices.data(); out->vertexRefCount = (uint32_t)res->vertices.size();
    out->triangles = res->triangles.data(); out->triangleBytes = (uint32_t)res->triangles.size();
    out->owner = res;
    return 0;
}
void clodref_dag_release(void*, brmi_dag* dag) { if (dag) { delete reinterpret_cast<Result*>(dag->owner); std::memset(dag, 0, sizeof(*dag)); } }
uint32_t clodref_meshoptimizer_version(void) { return MESHOPTIMIZER_VERSION; }

}  // extern "C"
