// brmi_passes.hpp -- C++ host mirror of the reference's pass interface for the visibility-buffer path.
//
// The reference schedules this path as OpenRenderGraph passes: subclasses of `ComputePass` with
//   DeclareResourceUsages(ComputePassBuilder*) / Setup() / Update(const UpdateExecutionContext&) /
//   Execute(PassExecutionContext&) / Cleanup()
// (e.g. BR/include/Render/GraphExtensions/ClusterLOD/ClusterSoftwareRasterizationPass.h:16-54,
//  BR/include/RenderPasses/DeferredShadingPass.h:11-121) spliced into the graph by
// `CLodExtension::GatherStructuralPasses` (BR/src/Render/GraphExtensions/CLodExtension.cpp:1411-2095)
// and `RenderGraphBuildHelper` (BR/include/Render/RenderGraphBuildHelper.h:220-414).
//
// OpenRenderGraph / BasicRHI sources are absent from the reference checkout (empty submodules), so the
// minimal surface those classes need is restated here under the same names and phase semantics; every
// pass forwards to the C ABI of libbrmi.so (include/brmi.h).  A maintainer with the real graph derives
// these classes from the real `ComputePass` instead of `brmi::host::ComputePass` and nothing else changes
// (INTEGRATION.md).  Failures throw std::runtime_error, as the reference's passes do
// (BR/src/Renderer.cpp:2112-2122).
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "brmi.h"

namespace brmi::host {

struct PassReturn {};                                   // reference: Execute() returns PassReturn{}
struct UpdateExecutionContext { const brmi_camera* mainCamera; const brmi_per_frame* perFrame; uint32_t frameIndex; };
struct PassExecutionContext { brmi_stream commandList; uint32_t frameIndex; float deltaTime; };   // commandList -> hipStream_t

// what DeclareResourceUsages reports: the reference's four builder verbs (census over the reference's passes: WithShaderResource 159,
// WithUnorderedAccess 143, WithConstantBuffer 55, WithIndirectArguments 23 uses), by Builtin:: resource name
struct ComputePassBuilder {
    std::vector<std::string> shaderResources, unorderedAccess, indirectArguments, constantBuffers;
    ComputePassBuilder& WithShaderResource(std::string n) { shaderResources.push_back(std::move(n)); return *this; }
    ComputePassBuilder& WithUnorderedAccess(std::string n) { unorderedAccess.push_back(std::move(n)); return *this; }
    // the reference's ExecuteIndirect argument / count buffers: here the counters of the workspace that size the device-side loops
    ComputePassBuilder& WithIndirectArguments(std::string n) { indirectArguments.push_back(std::move(n)); return *this; }
    ComputePassBuilder& WithConstantBuffer(std::string n) { constantBuffers.push_back(std::move(n)); return *this; }
    bool Mentions(const std::string& n) const {
        for (const auto* v : {&shaderResources, &unorderedAccess, &indirectArguments, &constantBuffers}) for (const auto& e : *v) if (e == n) return true;
        return false;
    }
};

// One shared brmi_pass per view; the individual passes are views on its stages (the reference shares the
// same buffers between these passes through the resource registry).
class PassState {
public:
    explicit PassState(const brmi_config& cfg) { check(brmi_create(&cfg, &pass_), "brmi_create"); }
    ~PassState() { brmi_destroy(pass_); }
    PassState(const PassState&) = delete;
    PassState& operator=(const PassState&) = delete;
    brmi_pass* get() const { return pass_; }
    void check(int rc, const char* what) const {
        if (rc != BRMI_OK) throw std::runtime_error(std::string(what) + ": " + (pass_ ? brmi_last_error(pass_) : "invalid argument"));
    }
    void SetScene(const brmi_scene_buffers& sc) { check(brmi_set_scene(pass_, &sc), "brmi_set_scene"); }
    std::vector<brmi_resource_desc> Declare() {
        std::vector<brmi_resource_desc> out;
        check(brmi_declare(pass_, [](void* u, const brmi_resource_desc* d) { static_cast<std::vector<brmi_resource_desc>*>(u)->push_back(*d); }, &out), "brmi_declare");
        return out;
    }
    void Bind(const std::vector<brmi_resource_binding>& b, brmi_stream s) { check(brmi_setup(pass_, b.data(), (uint32_t)b.size(), s), "brmi_setup"); }
    void Update(const UpdateExecutionContext& u, brmi_stream s) {
        brmi_frame_update f{u.mainCamera, u.perFrame, u.frameIndex};
        check(brmi_update(pass_, &f, s), "brmi_update");
    }
private:
    brmi_pass* pass_ = nullptr;
};

class ComputePass {
public:
    explicit ComputePass(std::shared_ptr<PassState> st, std::string name) : state(std::move(st)), name_(std::move(name)) {}
    virtual ~ComputePass() = default;
    virtual void DeclareResourceUsages(ComputePassBuilder* builder) = 0;
    virtual void Setup() {}
    virtual void Update(const UpdateExecutionContext&) {}
    virtual PassReturn Execute(PassExecutionContext& ctx) = 0;
    virtual void Cleanup() {}
    const std::string& Name() const { return name_; }
protected:
    std::shared_ptr<PassState> state;
    std::string name_;
};

#define BRMI_STAGE_PASS(CLASS, NAME, CALL, SRVS, UAVS)                                                        \
    class CLASS final : public ComputePass {                                                                  \
    public:                                                                                                   \
        explicit CLASS(std::shared_ptr<PassState> st) : ComputePass(std::move(st), NAME) {}                   \
        void DeclareResourceUsages(ComputePassBuilder* b) override {                                          \
            for (const char* s : std::vector<const char*> SRVS) b->WithShaderResource(s);                     \
            for (const char* u : std::vector<const char*> UAVS) b->WithUnorderedAccess(u);                    \
            b->WithConstantBuffer("Builtin::PerFrameBuffer");                                                 \
            b->WithIndirectArguments("brmi::Workspace");   /* cluster / record counts read on the device */   \
        }                                                                                                     \
        PassReturn Execute(PassExecutionContext& ctx) override { state->check(CALL, NAME); return {}; }       \
    };

// reference: ClearVisibilityBufferPass (BR/include/RenderPasses/ClearVisibilityBufferPass.h)
BRMI_STAGE_PASS(ClearVisibilityBufferPass, "ClearVisibilityBufferPass", brmi_clear_visibility(state->get(), ctx.commandList),
                ({}), ({"Builtin::PrimaryCamera::VisibilityTexture"}))
// reference: HierarchicalDispatchCullingPass phase 1 (BR/src/Render/GraphExtensions/ClusterLOD/HierarchicalDispatchCullingPass.cpp:496-1114)
BRMI_STAGE_PASS(HierarchicalCullingPass1, "HierarchicalCullingPass1", brmi_cull(state->get(), 1, ctx.commandList),
                ({"Builtin::PerMeshInstanceBuffer", "Builtin::PerObjectBuffer", "Builtin::CameraBuffer", "Builtin::CullingCameraBuffer", "Builtin::CLod::Nodes",
                  "Builtin::CLod::Groups", "Builtin::CLod::Segments", "Builtin::CLod::GroupPageMap", "Builtin::CLod::MeshMetadata", "Builtin::CLod::Offsets"}),
                ({"Builtin::CLod::VisibleClusters", "brmi::Workspace"}))
// reference: ClusterSoftwareRasterizationPass (BR/src/Render/GraphExtensions/ClusterLOD/ClusterSoftwareRasterizationPass.cpp:154-207);
// the raster-bucket histogram / scan / compaction passes (RasterBucket*Pass) collapse into it
BRMI_STAGE_PASS(SoftwareRasterizeClustersPass1, "SoftwareRasterizeClustersPass1", brmi_raster(state->get(), 1, ctx.commandList),
                ({"Builtin::CLod::VisibleClusters", "Builtin::PerMeshInstanceBuffer", "Builtin::PerObjectBuffer", "Builtin::CullingCameraBuffer", "CLod page slabs",
                  "Builtin::PerMaterialDataBuffer", "material textures + samplers (alpha test)"}),
                ({"Builtin::PrimaryCamera::VisibilityTexture"}))
// reference: PerViewLinearDepthCopyPass (BR/src/Render/GraphExtensions/ClusterLOD/PerViewLinearDepthCopyPass.cpp)
BRMI_STAGE_PASS(LinearDepthCopyPass1, "LinearDepthCopyPass1", brmi_depth_copy(state->get(), ctx.commandList),
                ({"Builtin::PrimaryCamera::VisibilityTexture"}), ({"Builtin::PrimaryCamera::LinearDepthMap"}))
// reference: linear-depth downsample (FidelityFX SPD, BR/include/RenderPasses/FidelityFX/Downsample.h; scheduled at CLodExtension.cpp:1920-1949)
BRMI_STAGE_PASS(LinearDepthDownsamplePass, "LinearDepthDownsamplePass", brmi_build_hzb(state->get(), ctx.commandList),
                ({"Builtin::PrimaryCamera::LinearDepthMap"}), ({"Builtin::PrimaryCamera::LinearDepthMap(mips)"}))
// reference: phase 2 of the occlusion chain (CLodExtension.cpp:2002-2088): replays what phase 1 found occluded
BRMI_STAGE_PASS(HierarchicalCullingPass2, "HierarchicalCullingPass2", brmi_cull(state->get(), 2, ctx.commandList),
                ({"Builtin::PrimaryCamera::LinearDepthMap(mips)", "Builtin::PerMeshInstanceBuffer", "Builtin::PerObjectBuffer", "Builtin::CameraBuffer", "brmi::Workspace"}),
                ({"Builtin::CLod::VisibleClusters", "brmi::Workspace"}))
BRMI_STAGE_PASS(SoftwareRasterizeClustersPass2, "SoftwareRasterizeClustersPass2", brmi_raster(state->get(), 2, ctx.commandList),
                ({"Builtin::CLod::VisibleClusters", "Builtin::PerMeshInstanceBuffer", "Builtin::PerObjectBuffer", "Builtin::CullingCameraBuffer", "CLod page slabs",
                  "Builtin::PerMaterialDataBuffer", "material textures + samplers (alpha test)"}),
                ({"Builtin::PrimaryCamera::VisibilityTexture"}))
// reference: MaterialHistogram .. EvaluateMaterialGroups (BR/include/RenderPasses/VisUtil/*.h; parameter list at EvaluateMaterialGroupsPass.h:68-111)
BRMI_STAGE_PASS(EvaluateMaterialGroupsPass, "EvaluateMaterialGroupsPass", brmi_gbuffer(state->get(), ctx.commandList),
                ({"Builtin::PrimaryCamera::VisibilityTexture", "Builtin::CLod::VisibleClusters", "Builtin::PerMaterialDataBuffer", "Builtin::PerMaterialOpenPBRDataBuffer",
                  "Builtin::NormalMatrixBuffer", "CLod page slabs", "material textures + samplers"}),
                ({"Builtin::GBuffer::Normals", "Builtin::GBuffer::Albedo", "Builtin::GBuffer::Coat", "Builtin::GBuffer::Emissive", "Builtin::GBuffer::Fuzz",
                  "Builtin::GBuffer::MetallicRoughness", "Builtin::GBuffer::MotionVectors", "Builtin::PrimaryCamera::LinearDepthMap"}))
// reference: ClusterGenerationPass + LightCullingPass (BR/include/RenderPasses/ClusterGenerationPass.h:41-42, LightCullingPass.h:49-51)
BRMI_STAGE_PASS(LightCullingPass, "LightCullingPass", brmi_light_clustering(state->get(), ctx.commandList),
                ({"Builtin::Light::InfoBuffer", "Builtin::Light::ActiveLightIndices", "Builtin::CameraBuffer"}),
                ({"Builtin::Light::ClusterBuffer", "Builtin::Light::PagesBuffer"}))
// reference: DeferredShadingPass (BR/include/RenderPasses/DeferredShadingPass.h:77-107)
BRMI_STAGE_PASS(DeferredShadingPass, "DeferredShadingPass", brmi_shade(state->get(), ctx.commandList),
                ({"Builtin::GBuffer::Normals", "Builtin::GBuffer::Albedo", "Builtin::GBuffer::Coat", "Builtin::GBuffer::Emissive", "Builtin::GBuffer::Fuzz",
                  "Builtin::GBuffer::MetallicRoughness", "Builtin::PrimaryCamera::LinearDepthMap", "Builtin::Light::ClusterBuffer", "Builtin::Light::PagesBuffer",
                  "Builtin::OpenPBR::*"}),
                ({"Builtin::Color::HDRColorTarget"}))
#undef BRMI_STAGE_PASS

// ---- the extension interface (BR/include/Render/GraphExtensions/CLodExtension.h:20-31) ------------------------------------------
// Stand-ins for the graph-side types the five hooks take (OpenRenderGraph is an empty submodule in the reference checkout).
struct ResourceRegistry;                                 // opaque: the graph's name -> backing map
// what the extension needs from the graph: memory for a declared resource (the graph owns backing memory, aliasing and barriers:
// BR/src/Renderer.cpp:2536-2571) and the stream the frame is recorded on
struct RenderGraph {
    std::function<void*(const brmi_resource_desc&)> allocate;      // RegisterResource + backing
    brmi_stream stream = nullptr;
    ResourceRegistry* registry = nullptr;
};
struct ExternalInsertPoint {
    enum Kind { Begin, End, After, Before, Between } kind = End;
    std::string anchor, anchor2;
    static ExternalInsertPoint AtBegin() { return {Begin, "", ""}; }
    static ExternalInsertPoint AtEnd() { return {End, "", ""}; }
    static ExternalInsertPoint AfterPass(std::string a) { return {After, std::move(a), ""}; }
    static ExternalInsertPoint BeforePass(std::string a) { return {Before, std::move(a), ""}; }
    static ExternalInsertPoint BetweenPasses(std::string a, std::string b) { return {Between, std::move(a), std::move(b)}; }
};
struct ExternalPassDesc {                                // ExternalPassDesc::Compute(name, pass).At(point)
    std::string name; std::shared_ptr<ComputePass> pass; ExternalInsertPoint where; std::vector<std::string> alsoBefore;
    static ExternalPassDesc Compute(std::string n, std::shared_ptr<ComputePass> p) { ExternalPassDesc d; d.name = std::move(n); d.pass = std::move(p); return d; }
    ExternalPassDesc& At(ExternalInsertPoint p) { where = std::move(p); return *this; }
    ExternalPassDesc& AlsoBefore(std::string n) { alsoBefore.push_back(std::move(n)); return *this; }
};
class IRenderGraphExtension {
public:
    virtual ~IRenderGraphExtension() = default;
    virtual void PrepareForBuild(RenderGraph& rg) = 0;
    virtual void Initialize(RenderGraph& rg) = 0;
    virtual void OnRegistryReset(ResourceRegistry* reg) = 0;
    virtual void GatherStructuralPasses(RenderGraph& rg, std::vector<ExternalPassDesc>& outPasses) = 0;
    virtual void GatherFramePasses(RenderGraph& rg, std::vector<ExternalPassDesc>& outPasses) = 0;
};

// reference: CLodExtension -- the five hooks over one shared brmi_pass
class BrmiGraphExtension final : public IRenderGraphExtension {
public:
    explicit BrmiGraphExtension(std::shared_ptr<PassState> st, bool occlusionCulling = false) : state_(std::move(st)), occlusion_(occlusionCulling) {}
    // PrepareForBuild (CLodExtension.cpp: capacity / settings refresh before the graph is compiled): what the pass will ask the graph for
    void PrepareForBuild(RenderGraph&) override { declared_ = state_->Declare(); }
    // Initialize (InitializeCoreResources + RegisterResource): the graph allocates every declared resource, the pass binds them
    void Initialize(RenderGraph& rg) override {
        if (declared_.empty()) declared_ = state_->Declare();
        bindings_.clear();
        for (const brmi_resource_desc& d : declared_) {
            brmi_resource_desc padded = d; if (padded.bytes < 16) padded.bytes = 16;
            void* p = rg.allocate ? rg.allocate(padded) : nullptr;
            if (!p) throw std::runtime_error(std::string("BrmiGraphExtension::Initialize: the graph returned no backing for ") + d.name);
            bindings_.push_back({d.id, p, padded.bytes});
        }
        state_->Bind(bindings_, rg.stream);
        bound_ = true;
    }
    // OnRegistryReset: the graph dropped its backings (resize, device reset): nothing may be executed until Initialize ran again
    void OnRegistryReset(ResourceRegistry*) override { bound_ = false; bindings_.clear(); brmi_invalidate_hzb(state_->get()); }
    // GatherStructuralPasses: cull/raster chain spliced before "MaterialHistogramPass" (CLodExtension.cpp:1704,1910); with
    // enableOcclusionCulling the depth copy / downsample / phase-2 passes follow phase 1 and the chain is rebuilt from the final
    // depth for the next frame (CLodExtension.cpp:1920-2088)
    void GatherStructuralPasses(RenderGraph&, std::vector<ExternalPassDesc>& out) override {
        if (!bound_) throw std::runtime_error("BrmiGraphExtension::GatherStructuralPasses: Initialize has not run since the last registry reset");
        std::string prev;
        for (auto& p : GatherStructuralPasses()) {
            ExternalPassDesc d = ExternalPassDesc::Compute(p->Name(), p);
            d.At(prev.empty() ? ExternalInsertPoint::AtBegin() : ExternalInsertPoint::AfterPass(prev));
            if (p->Name() == "SoftwareRasterizeClustersPass1" || p->Name() == "SoftwareRasterizeClustersPass2") d.AlsoBefore("MaterialHistogramPass");
            prev = p->Name();
            out.push_back(std::move(d));
        }
    }
    // GatherFramePasses: the reference adds its per-frame streaming / telemetry passes here; this path has none (everything resident)
    void GatherFramePasses(RenderGraph&, std::vector<ExternalPassDesc>&) override {}
    const std::vector<brmi_resource_desc>& Declared() const { return declared_; }
    const std::vector<brmi_resource_binding>& Bindings() const { return bindings_; }
    // the plain pass list (order = the order the reference graph runs them)
    std::vector<std::shared_ptr<ComputePass>> GatherStructuralPasses() const {
        std::vector<std::shared_ptr<ComputePass>> p{std::make_shared<ClearVisibilityBufferPass>(state_), std::make_shared<HierarchicalCullingPass1>(state_),
                                                    std::make_shared<SoftwareRasterizeClustersPass1>(state_)};
        if (occlusion_) {
            p.push_back(std::make_shared<LinearDepthCopyPass1>(state_)); p.push_back(std::make_shared<LinearDepthDownsamplePass>(state_));
            p.push_back(std::make_shared<HierarchicalCullingPass2>(state_)); p.push_back(std::make_shared<SoftwareRasterizeClustersPass2>(state_));
        }
        p.push_back(std::make_shared<EvaluateMaterialGroupsPass>(state_));
        if (occlusion_) p.push_back(std::make_shared<LinearDepthDownsamplePass>(state_));
        p.push_back(std::make_shared<LightCullingPass>(state_)); p.push_back(std::make_shared<DeferredShadingPass>(state_));
        return p;
    }
private:
    std::shared_ptr<PassState> state_;
    bool occlusion_ = false, bound_ = false;
    std::vector<brmi_resource_desc> declared_;
    std::vector<brmi_resource_binding> bindings_;
};

}  // namespace brmi::host
