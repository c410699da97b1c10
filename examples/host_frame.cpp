// host_frame.cpp -- a C++ host driving the path exactly as a render graph would: providers resolved to device
// buffers, resources declared by the pass and allocated by the "graph" (here: hipMalloc), passes gathered from the
// extension and executed in order on one stream.  Prints the frame's checksums as one JSON line.
//
//   build:  make host_example        run:  basicrenderer_amd/lib/brmi_host_frame [preset W H lights occlusion frames materialFeatures lodLevels framesInFlight]
// framesInFlight = 2 (needs occlusion = 1): after the graph-driven frames, the same frames again through two linked passes that alternate on a
// geometry stream and a shading stream (brmi_set_history_source + brmi_execute_split); the line then reports the last of THOSE frames.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../basicrenderer_amd/host/brmi_passes.hpp"
#include "brmi_scene.h"

#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(2); } } while (0)

static uint64_t fnv1a(const void* p, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(p); uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

template <typename T> static const T* upload(const brmi_scene* sc, uint32_t id, uint32_t* count, std::vector<void*>& keep) {
    const void* p; uint64_t bytes; uint32_t n;
    if (brmi_scene_array(sc, id, &p, &bytes, &n) != 0) std::exit(3);
    if (count) *count = n;
    if (bytes == 0) return nullptr;
    void* d; HIPCHK(hipMalloc(&d, bytes)); HIPCHK(hipMemcpy(d, p, bytes, hipMemcpyHostToDevice)); keep.push_back(d);
    return static_cast<const T*>(d);
}

int main(int argc, char** argv) {
    using namespace brmi::host;
    brmi_scene_params prm{};
    prm.preset = argc > 1 ? (uint32_t)std::atoi(argv[1]) : BRMI_PRESET_TINY;
    prm.width = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 256; prm.height = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 144;
    prm.numPointLights = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 6; prm.withDirectionalLight = 1; prm.sizeScale = 1.0f;
    const bool occlusion = argc > 5 && std::atoi(argv[5]) != 0;      // 2-phase HZB occlusion culling: the unfused pass sequence of the reference graph
    const int frames = argc > 6 ? std::max(1, std::atoi(argv[6])) : 1;
    prm.materialFeatures = argc > 7 ? (uint32_t)std::atoi(argv[7]) : 0u;        // brmi_scene.h: 8 = texture-sampled, 16 = alpha-tested materials
    prm.lodLevels = argc > 8 ? (uint32_t)std::atoi(argv[8]) : 0u;
    const int framesInFlight = argc > 9 ? std::atoi(argv[9]) : 1;
    brmi_scene* scene = nullptr;
    if (prm.preset == 100u) {
        // preset 100: not a preset -- the host's OWN geometry through brmi_scene_create_from_meshes (row f-1).  A height field whose coordinates
        // are exact binary fractions (the Python test builds the same arrays), drawn twice; normals are left to the library.
        const uint32_t n = 48;
        std::vector<float> pos, uv; std::vector<uint32_t> idx;
        for (uint32_t i = 0; i <= n; i++) for (uint32_t j = 0; j <= n; j++) {
            pos.push_back((float)i * 0.125f - 3.0f); pos.push_back((float)((i * 7u + j * 13u) % 16u) / 64.0f - 0.5f); pos.push_back((float)j * 0.125f - 3.0f);
            uv.push_back((float)i / 8.0f); uv.push_back((float)j / 8.0f);
        }
        for (uint32_t i = 0; i < n; i++) for (uint32_t j = 0; j < n; j++) {
            const uint32_t a = i * (n + 1) + j, b = a + (n + 1), c = b + 1, d = a + 1;
            const uint32_t q[6] = {a, d, c, a, c, b};
            idx.insert(idx.end(), q, q + 6);
        }
        brmi_mesh_input mesh{}; mesh.positions = pos.data(); mesh.uvs = uv.data(); mesh.vertexCount = pos.size() / 3; mesh.indices = idx.data(); mesh.indexCount = idx.size(); mesh.material = 0;
        brmi_instance_input inst[2] = {};
        for (int k = 0; k < 2; k++) { inst[k].mesh = 0; for (int r = 0; r < 4; r++) inst[k].model[r][r] = 1.0f; }
        inst[1].model[3][0] = 1.5f; inst[1].model[3][1] = 0.75f; inst[1].model[3][2] = -2.0f;
        brmi_view_input view{}; view.eye[0] = 0.25f; view.eye[1] = 1.5f; view.eye[2] = 4.0f; view.yaw = 0.0f; view.pitch = -0.25f; view.fovYDegrees = 60.0f; view.zNear = 0.125f; view.zFar = 256.0f;
        scene = brmi_scene_create_from_meshes(&prm, &mesh, 1, inst, 2, &view, nullptr, nullptr, nullptr);
    } else scene = brmi_scene_create(&prm);
    if (!scene) return 1;
    std::vector<void*> keep;
    brmi_scene_buffers sb{};
    // slabs: device array of device pointers, entry 0 = "not resident"
    const uint32_t nslabs = brmi_scene_slab_count(scene);
    std::vector<const uint8_t*> slabPtrs(nslabs + 1, nullptr);
    for (uint32_t s = 1; s <= nslabs; s++) {
        const void* p; uint64_t bytes; brmi_scene_slab(scene, s, &p, &bytes);
        void* d; HIPCHK(hipMalloc(&d, bytes)); HIPCHK(hipMemcpy(d, p, bytes, hipMemcpyHostToDevice)); keep.push_back(d); slabPtrs[s] = static_cast<const uint8_t*>(d);
    }
    { void* d; HIPCHK(hipMalloc(&d, slabPtrs.size() * 8)); HIPCHK(hipMemcpy(d, slabPtrs.data(), slabPtrs.size() * 8, hipMemcpyHostToDevice)); keep.push_back(d);
      sb.slabs = static_cast<const uint8_t* const*>(d); sb.slabCount = nslabs + 1; }
    sb.perObject = upload<brmi_per_object>(scene, BRMI_ARR_PER_OBJECT, &sb.perObjectCount, keep);
    sb.normalMatrices = upload<float>(scene, BRMI_ARR_NORMAL_MATRICES, nullptr, keep);
    sb.perMesh = upload<brmi_per_mesh>(scene, BRMI_ARR_PER_MESH, &sb.perMeshCount, keep);
    sb.perMeshInstance = upload<brmi_per_mesh_instance>(scene, BRMI_ARR_PER_MESH_INSTANCE, &sb.perMeshInstanceCount, keep);
    sb.clodOffsets = upload<brmi_mesh_instance_clod_offsets>(scene, BRMI_ARR_CLOD_OFFSETS, nullptr, keep);
    sb.meshMetadata = upload<brmi_clod_mesh_metadata>(scene, BRMI_ARR_CLOD_MESH_METADATA, &sb.meshMetadataCount, keep);
    sb.lodNodes = upload<brmi_lod_node>(scene, BRMI_ARR_LOD_NODES, &sb.lodNodeCount, keep);
    sb.lodGroups = upload<brmi_lod_group>(scene, BRMI_ARR_LOD_GROUPS, &sb.lodGroupCount, keep);
    sb.lodSegments = upload<brmi_lod_segment>(scene, BRMI_ARR_LOD_SEGMENTS, &sb.lodSegmentCount, keep);
    sb.groupPageMap = upload<brmi_group_page_map_entry>(scene, BRMI_ARR_GROUP_PAGE_MAP, &sb.groupPageMapCount, keep);
    sb.materials = upload<brmi_material_info>(scene, BRMI_ARR_MATERIALS, &sb.materialCount, keep);
    sb.openpbrMaterials = upload<brmi_openpbr_material_info>(scene, BRMI_ARR_OPENPBR_MATERIALS, &sb.openpbrMaterialCount, keep);
    sb.lights = upload<brmi_light_info>(scene, BRMI_ARR_LIGHTS, &sb.lightCount, keep);
    sb.activeLightIndices = upload<uint32_t>(scene, BRMI_ARR_ACTIVE_LIGHT_INDICES, nullptr, keep);
    sb.cameras = upload<brmi_camera>(scene, BRMI_ARR_CAMERAS, &sb.cameraCount, keep);
    sb.cullingCameras = upload<brmi_culling_camera>(scene, BRMI_ARR_CULLING_CAMERAS, nullptr, keep);
    sb.viewRasterInfo = upload<brmi_view_raster_info>(scene, BRMI_ARR_VIEW_RASTER_INFO, nullptr, keep);
    sb.perFrame = upload<brmi_per_frame>(scene, BRMI_ARR_PER_FRAME, nullptr, keep);
    sb.activeDraws = upload<uint32_t>(scene, BRMI_ARR_ACTIVE_DRAWS, &sb.activeDrawCount, keep);
    sb.skinningMatrices = upload<float>(scene, BRMI_ARR_SKINNING_MATRICES, &sb.skinningMatrixCount, keep);
    sb.lutOpaqueDielectricEnergyComplement = upload<uint16_t>(scene, BRMI_ARR_LUT_OD_ENERGY, nullptr, keep);
    sb.lutOpaqueDielectricAvgEnergyComplement = upload<uint16_t>(scene, BRMI_ARR_LUT_OD_AVG_ENERGY, nullptr, keep);
    sb.lutIdealMetalEnergyComplement = upload<uint16_t>(scene, BRMI_ARR_LUT_IM_ENERGY, nullptr, keep);
    sb.lutIdealMetalAvgEnergyComplement = upload<uint16_t>(scene, BRMI_ARR_LUT_IM_AVG_ENERGY, nullptr, keep);
    sb.lutFuzzLTC = upload<float>(scene, BRMI_ARR_LUT_FUZZ_LTC, nullptr, keep);
    {   // material textures: the generator's descriptors hold byte offsets into the texel array; the host relocates them to device pointers
        const void* hp; uint64_t bytes; uint32_t n;
        brmi_scene_array(scene, BRMI_ARR_TEXTURE_DESCS, &hp, &bytes, &n);
        if (n) {
            const uint8_t* texels = upload<uint8_t>(scene, BRMI_ARR_TEXELS, nullptr, keep);
            std::vector<brmi_texture_desc> descs(static_cast<const brmi_texture_desc*>(hp), static_cast<const brmi_texture_desc*>(hp) + n);
            for (auto& d : descs) d.texels = texels + reinterpret_cast<uintptr_t>(d.texels);
            void* dd; HIPCHK(hipMalloc(&dd, bytes)); HIPCHK(hipMemcpy(dd, descs.data(), bytes, hipMemcpyHostToDevice)); keep.push_back(dd);
            sb.textures = static_cast<const brmi_texture_desc*>(dd); sb.textureCount = n;
            sb.samplers = upload<brmi_sampler_desc>(scene, BRMI_ARR_SAMPLER_DESCS, &sb.samplerCount, keep);
            sb.srgbToLinear = upload<float>(scene, BRMI_ARR_SRGB_TO_LINEAR, nullptr, keep);
        }
    }

    hipStream_t stream; HIPCHK(hipStreamCreate(&stream));
    try {
        brmi_config cfg; brmi_default_config(&cfg, prm.width, prm.height);
        cfg.maxVisibleClusters = 1u << 16; cfg.maxTraversalRecords = 1u << 16; cfg.enableOcclusionCulling = occlusion ? 1u : 0u;
        auto state = std::make_shared<PassState>(cfg);
        state->SetScene(sb);
        // the five hooks of the reference's IRenderGraphExtension, in the order the graph calls them; the "graph" owns the memory of
        // every declared resource (here: hipMalloc)
        std::vector<uint64_t> sizes(BRMI_RES_COUNT, 0);
        RenderGraph rg;
        rg.stream = stream;
        rg.allocate = [&](const brmi_resource_desc& d) { void* p = nullptr; HIPCHK(hipMalloc(&p, d.bytes)); HIPCHK(hipMemset(p, 0, d.bytes)); keep.push_back(p); return p; };
        BrmiGraphExtension ext(state, occlusion);
        ext.PrepareForBuild(rg);
        for (const brmi_resource_desc& d : ext.Declared()) sizes[d.id] = d.bytes;
        ext.Initialize(rg);
        ext.OnRegistryReset(rg.registry);           // a registry reset (resize): the passes are unusable until Initialize ran again
        bool refused = false;
        { std::vector<ExternalPassDesc> none; try { ext.GatherStructuralPasses(rg, none); } catch (const std::runtime_error&) { refused = true; } }
        if (!refused) throw std::runtime_error("GatherStructuralPasses ran on a reset registry");
        ext.Initialize(rg);
        const std::vector<brmi_resource_binding> binds = ext.Bindings();
        std::vector<ExternalPassDesc> descs, framePasses;
        ext.GatherStructuralPasses(rg, descs);
        ext.GatherFramePasses(rg, framePasses);
        std::vector<std::shared_ptr<ComputePass>> passes;
        for (auto& d : descs) passes.push_back(d.pass);
        ComputePassBuilder builder;
        for (auto& p : passes) { p->DeclareResourceUsages(&builder); p->Setup(); }
        // every resource the C ABI declares is named by some pass of the chain (the HZB chain only with occlusion culling: its passes are not scheduled without)
        for (const brmi_resource_desc& d : ext.Declared())
            if (!builder.Mentions(d.name) && !(d.id == BRMI_RES_HZB && !occlusion) && !(d.id == BRMI_RES_LINEAR_DEPTH && false))
                throw std::runtime_error(std::string("no pass declares ") + d.name);
        const void* camHost; const void* pfHost; uint64_t b; uint32_t n;
        brmi_scene_array(scene, BRMI_ARR_CAMERAS, &camHost, &b, &n); brmi_scene_array(scene, BRMI_ARR_PER_FRAME, &pfHost, &b, &n);
        state->Update({static_cast<const brmi_camera*>(camHost), static_cast<const brmi_per_frame*>(pfHost), 0}, stream);
        for (int f = 0; f < frames; f++) {
            PassExecutionContext ctx{stream, (uint32_t)f, 0.0f};
            for (auto& p : passes) p->Execute(ctx);
        }
        HIPCHK(hipStreamSynchronize(stream));
        std::vector<brmi_resource_binding> lastBinds = binds;
        std::shared_ptr<PassState> last = state;
        std::shared_ptr<PassState> second;
        if (framesInFlight == 2) {
            if (!occlusion) throw std::runtime_error("framesInFlight = 2 needs occlusion = 1 (there is no history to share otherwise)");
            // the second pass owns a full set of resources; its phase 1 reads the first pass's depth chain and vice versa
            second = std::make_shared<PassState>(cfg);
            // per-frame inputs are per pass (include/brmi.h, brmi_execute_split): the second pass gets its own camera, culling camera, raster
            // info and per-frame record; geometry, materials and lights are read-only and shared
            brmi_scene_buffers sb2 = sb;
            sb2.cameras = upload<brmi_camera>(scene, BRMI_ARR_CAMERAS, &sb2.cameraCount, keep);
            sb2.cullingCameras = upload<brmi_culling_camera>(scene, BRMI_ARR_CULLING_CAMERAS, nullptr, keep);
            sb2.viewRasterInfo = upload<brmi_view_raster_info>(scene, BRMI_ARR_VIEW_RASTER_INFO, nullptr, keep);
            sb2.perFrame = upload<brmi_per_frame>(scene, BRMI_ARR_PER_FRAME, nullptr, keep);
            second->SetScene(sb2);
            std::vector<brmi_resource_binding> binds2;
            // at least 16 bytes each, like BrmiGraphExtension::Initialize's bindings
            for (const brmi_resource_desc& d : second->Declare()) { brmi_resource_desc pd = d; if (pd.bytes < 16) pd.bytes = 16; binds2.push_back(brmi_resource_binding{d.id, rg.allocate(pd), pd.bytes}); }
            second->Bind(binds2, stream);
            second->Update({static_cast<const brmi_camera*>(camHost), static_cast<const brmi_per_frame*>(pfHost), 0}, stream);
            state->check(brmi_invalidate_hzb(state->get()), "brmi_invalidate_hzb");       // start the sequence over: frame 0 has no history
            state->check(brmi_set_history_source(state->get(), second->get()), "brmi_set_history_source");
            second->check(brmi_set_history_source(second->get(), state->get()), "brmi_set_history_source");
            int least = 0, greatest = 0; HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            hipStream_t geometry, shading;
            HIPCHK(hipStreamCreateWithPriority(&geometry, hipStreamNonBlocking, greatest)); HIPCHK(hipStreamCreateWithPriority(&shading, hipStreamNonBlocking, least));
            HIPCHK(hipStreamSynchronize(stream));
            PassState* pair[2] = {state.get(), second.get()};
            for (int f = 0; f < frames; f++) pair[f & 1]->check(brmi_execute_split(pair[f & 1]->get(), geometry, shading), "brmi_execute_split");
            HIPCHK(hipStreamSynchronize(shading)); HIPCHK(hipStreamSynchronize(geometry));
            HIPCHK(hipStreamDestroy(geometry)); HIPCHK(hipStreamDestroy(shading));
            if ((frames - 1) & 1) { last = second; lastBinds = binds2; }
        }
        brmi_counters c; last->check(brmi_read_counters(last->get(), &c, stream), "brmi_read_counters");
        auto checksum = [&](uint32_t id) { std::vector<uint8_t> h(sizes[id]); for (auto& bd : lastBinds) if (bd.id == id) HIPCHK(hipMemcpy(h.data(), bd.ptr, sizes[id], hipMemcpyDeviceToHost)); return fnv1a(h.data(), h.size()); };
        std::printf("{\"passes\": %zu, \"frame_passes\": %zu, \"srv\": %zu, \"uav\": %zu, \"cbv\": %zu, \"indirect\": %zu, \"visible_clusters\": %u, \"visible_clusters_phase2\": %u, \"replayed\": %u, \"vis_fnv\": \"%016llx\", \"hdr_fnv\": \"%016llx\", \"normals_fnv\": \"%016llx\"}\n",
                    passes.size(), framePasses.size(), builder.shaderResources.size(), builder.unorderedAccess.size(), builder.constantBuffers.size(), builder.indirectArguments.size(), c.visibleClusters, c.visibleClustersPhase2, c.replayNodes + c.replayMeshlets,
                    (unsigned long long)checksum(BRMI_RES_VISIBILITY), (unsigned long long)checksum(BRMI_RES_HDR_COLOR), (unsigned long long)checksum(BRMI_RES_GBUF_NORMALS));
    } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 4; }
    for (void* p : keep) (void)hipFree(p);
    brmi_scene_destroy(scene);
    return 0;
}
