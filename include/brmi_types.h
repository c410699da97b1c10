/*
 * brmi_types.h -- the data contract of the visibility-buffer hot path.
 *
 * Plain-C POD mirrors of the GPU-visible structures BasicRenderer's shaders index.
 * Every struct is byte-for-byte the layout the reference uploads, so buffers produced by
 * the reference's managers could be handed to this library unchanged.  Citations are
 * `path:line` relative to the reference checkout (BR/ = BasicRenderer/).
 *
 * Matrices are row-major float[4][4]; vectors multiply from the left (row-vector
 * convention, HLSL `mul(v, M)`), exactly as the reference (`row_major matrix`).
 */
#ifndef BRMI_TYPES_H
#define BRMI_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants -------------------------------------------------------------------------- */
/* BR/shaders/Common/defines.h:3-11 */
#define BRMI_MESHLET_MAX_VERTS        128u
#define BRMI_MESHLET_MAX_TRIS         128u
#define BRMI_PAGE_SIZE                (256u * 1024u)
/* BR/shaders/Include/clodStructs.hlsli:36-44 */
#define BRMI_PAGE_ATTRIBUTE_NORMAL    (1u << 0)
#define BRMI_PAGE_ATTRIBUTE_JOINTS    (1u << 1)
#define BRMI_PAGE_ATTRIBUTE_WEIGHTS   (1u << 2)
#define BRMI_PAGE_ATTRIBUTE_COLOR     (1u << 3)
#define BRMI_POSITION_FORMAT_FLOAT3   1u
#define BRMI_POSITION_STRIDE_BYTES    12u
/* BR/shaders/Include/clodStructs.hlsli:189-192 */
#define BRMI_NODE_INTERNAL            0u
#define BRMI_NODE_SEGMENT_LEAF        2u
#define BRMI_GROUP_FLAG_IS_VOXEL      (1u << 0)
/* BR/shaders/ClusterLOD/workGraphCulling.hlsl:1021 */
#define BRMI_BVH_MAX_CHILDREN         8u
/* BR/shaders/Include/visibilityPacking.hlsli:7-9 */
#define BRMI_VIS_TRI_BITS             7u
#define BRMI_VIS_CLUSTER_BITS         26u
#define BRMI_VIS_META_BITS            33u
#define BRMI_VIS_EMPTY                0xFFFFFFFFFFFFFFFFull
#define BRMI_DEPTH_EMPTY_BITS         0x7F7FFFFFu   /* BR/shaders/gbuffer.hlsl:132 */
/* BR/shaders/Include/structs.hlsli:517-518 */
#define BRMI_LIGHTS_PER_PAGE          12u
#define BRMI_LIGHT_PAGE_NULL          0xFFFFFFFFu
/* BR/src/Managers/LightManager.cpp:55-58 */
#define BRMI_LIGHT_PAGES_PER_CLUSTER  10u
/* light types, BR/shaders/Include/lighting.hlsli:89-106 */
#define BRMI_LIGHT_POINT              0u
#define BRMI_LIGHT_SPOT               1u
#define BRMI_LIGHT_DIRECTIONAL        2u
/* BR/include/ShaderBuffers.h:101, BR/shaders/Include/vertexFlags.hlsli */
#define BRMI_OBJECT_FLAG_REVERSE_WINDING (1u << 0)
#define BRMI_VERTEX_SKINNED           (1u << 3)
/* BR/shaders/Include/materialFlags.hlsli (subset the path reads) */
#define BRMI_MATERIAL_TEXTURED             (1u << 0)
#define BRMI_MATERIAL_BASE_COLOR_TEXTURE   (1u << 1)
#define BRMI_MATERIAL_NORMAL_MAP           (1u << 2)
#define BRMI_MATERIAL_AO_TEXTURE           (1u << 3)
#define BRMI_MATERIAL_EMISSIVE_TEXTURE     (1u << 4)
#define BRMI_MATERIAL_METALLIC_TEXTURE     (1u << 6)
#define BRMI_MATERIAL_ROUGHNESS_TEXTURE    (1u << 7)
#define BRMI_MATERIAL_DOUBLE_SIDED         (1u << 8)
#define BRMI_MATERIAL_PARALLAX             (1u << 9)    /* contact-refinement parallax of the height map: moves the texcoord of the slots below */
#define BRMI_MATERIAL_NEGATE_NORMALS       (1u << 10)
#define BRMI_MATERIAL_INVERT_NORMAL_GREEN  (1u << 11)
#define BRMI_MATERIAL_OPACITY_TEXTURE      (1u << 12)
#define BRMI_MATERIAL_ALPHA_TEST           (1u << 13)
/* every material texture slot whose sample reaches the G-buffer (the height map only moves their texcoord; the OpenPBR coat / fuzz slots live in textureBindings) */
#define BRMI_MATERIAL_ANY_TEXTURE (BRMI_MATERIAL_BASE_COLOR_TEXTURE | BRMI_MATERIAL_NORMAL_MAP | BRMI_MATERIAL_AO_TEXTURE | \
                                   BRMI_MATERIAL_EMISSIVE_TEXTURE | BRMI_MATERIAL_METALLIC_TEXTURE | BRMI_MATERIAL_ROUGHNESS_TEXTURE | \
                                   BRMI_MATERIAL_OPACITY_TEXTURE)
/* page attribute: the page carries UV sets (CLodPageHeader::uvSetCount > 0) */
#define BRMI_UV_QUANTIZATION_SCALE    65535.0f      /* BR/src/Mesh/ClusterLODUtilities.cpp:45-46 */
/* BR/shaders/Include/constants.hlsli:9-12 */
#define BRMI_MIN_PERCEPTUAL_ROUGHNESS 0.06f
#define BRMI_MIN_N_DOT_V              1e-4f
/* OpenPBR lookup tables, BR/shaders/Include/IBL.hlsli:307-311 */
#define BRMI_OPENPBR_TABLE_SIZE       32u

/* ---- page slab contents ------------------------------------------------------------------ */
/* BR/shaders/Include/clodStructs.hlsli:48-66 ; BR/include/Mesh/ClusterLODShaderTypes.h:26-47 */
typedef struct brmi_page_header {
    uint32_t meshletCount;
    uint32_t compressedPositionQuantExp;   /* BRMI_POSITION_FORMAT_* */
    uint32_t attributeMask;
    uint32_t uvSetCount;
    uint32_t descriptorOffset;
    uint32_t uvDescriptorOffset;
    uint32_t positionBitstreamOffset;
    uint32_t normalArrayOffset;
    uint32_t colorArrayOffset;
    uint32_t jointArrayOffset;
    uint32_t weightArrayOffset;
    uint32_t uvBitstreamDirectoryOffset;
    uint32_t triangleStreamOffset;
    uint32_t boneIndexStreamOffset;
    uint32_t reserved0;
    uint32_t reserved1;
} brmi_page_header;                         /* 64 B */

/* BR/shaders/Include/clodStructs.hlsli:70-89 */
typedef struct brmi_meshlet_descriptor {
    uint32_t positionBitOffset;             /* byte offset into the page position stream */
    uint32_t vertexAttributeOffset;         /* element offset into per-vertex attribute arrays */
    uint32_t triangleByteOffset;
    uint32_t boneListOffset;
    int32_t  minQx, minQy, minQz;
    uint32_t bitsAndVertexCount;            /* reserved:24 | vertexCount:8 (<<24) */
    uint32_t triangleCountAndRefinedGroup;  /* triangleCount:16 | (refinedGroup+1):16 */
    uint32_t boneCount;
    uint32_t sourceGroupLocalIndex;
    uint32_t reserved3;
    float    bounds[4];                     /* sphere cx,cy,cz,r (mesh space) */
} brmi_meshlet_descriptor;                  /* 64 B */

/* BR/shaders/Include/clodStructs.hlsli:92-102 */
typedef struct brmi_meshlet_uv_descriptor {
    uint32_t uvBitOffset;
    float    uvMinU, uvMinV, uvScaleU, uvScaleV;
    uint32_t uvBits;                        /* bitsU:8 | bitsV:8 */
    uint32_t reserved0, reserved1;
} brmi_meshlet_uv_descriptor;               /* 32 B */

/* ---- cluster-LOD hierarchy --------------------------------------------------------------- */
/* BR/shaders/ClusterLOD/workGraphCulling.hlsl:63-85 */
typedef struct brmi_lod_node {
    uint32_t isLeaf;          /* 0 internal, 2 segment leaf */
    uint32_t indexOrOffset;   /* internal: first child (rel. lodNodesBase); leaf: mesh-local segment index */
    uint32_t countMinusOne;   /* internal: childCount-1; leaf: refinedGroup+1 (0 = terminal) */
    uint32_t ownerGroupId;    /* leaf: mesh-local group index */
    float    cullCenterAndRadius[4];
    float    lodCenterAndRadius[4];
    float    maxQuadricError;
    float    pad0[3];
} brmi_lod_node;                            /* 64 B */

/* BR/shaders/Include/clodStructs.hlsli:153-187 */
typedef struct brmi_lod_segment {
    int32_t  refinedGroup;        /* -1 => terminal */
    uint32_t firstMeshletInPage;
    uint32_t meshletCount;
    uint32_t pageIndex;           /* mesh-local page-map index */
} brmi_lod_segment;                         /* 16 B */

typedef struct brmi_lod_group {
    float    centerAndRadius[4];
    float    error;
    uint32_t firstMeshlet;
    uint32_t meshletCount;
    int32_t  depth;
    uint32_t firstGroupVertex;
    uint32_t groupVertexCount;
    uint32_t firstSegment;
    uint32_t segmentCount;
    uint32_t terminalSegmentCount;
    uint32_t flags;
    uint32_t pageMapBase;
    uint32_t pageCount;
    int32_t  parentGroupId;
    float    maxParentError;
    float    representationError;
} brmi_lod_group;                           /* 76 B */

/* BR/shaders/Include/clodStructs.hlsli:132-136 */
typedef struct brmi_group_page_map_entry {
    uint32_t slabDescriptorIndex;   /* 0 = not resident; we index a slab pointer table with it */
    uint32_t slabByteOffset;
} brmi_group_page_map_entry;

/* BR/shaders/Include/clodStructs.hlsli:4-22 */
typedef struct brmi_mesh_instance_clod_offsets { uint32_t clodMeshMetadataIndex; } brmi_mesh_instance_clod_offsets;
typedef struct brmi_clod_mesh_metadata {
    uint32_t groupsBase;
    uint32_t segmentsBase;
    uint32_t lodNodesBase;
    uint32_t rootNode;
    uint32_t groupChunkTableBase;
    uint32_t groupChunkTableCount;
    uint32_t pageMapBase;
    uint32_t lodLevelInfoBase;
    uint32_t lodLevelCount;
    uint32_t maxDepth;
} brmi_clod_mesh_metadata;                  /* 40 B */

/* ---- per-object / per-mesh --------------------------------------------------------------- */
/* BR/shaders/Include/structs.hlsli:497-531 ; BR/include/ShaderBuffers.h:103-136 */
typedef struct brmi_per_object {
    float    model[4][4];
    float    prevModel[4][4];
    float    modelInverse[4][4];
    uint32_t normalMatrixBufferIndex;
    uint32_t objectFlags;
    uint32_t pad[2];
} brmi_per_object;                          /* 208 B */

typedef struct brmi_per_mesh {
    uint32_t materialDataIndex;
    uint32_t rasterBucketIndex;
    uint32_t vertexFlags;
    uint32_t vertexByteSize;
    uint32_t skinningVertexByteSize;
    float    boundingSphere[4];
    uint32_t clodMeshletBufferOffset;
    uint32_t clodMeshletVerticesBufferOffset;
    uint32_t clodMeshletTrianglesBufferOffset;
    uint32_t clodNumMeshlets;
    uint32_t vertexBufferOffset;
    uint32_t numVertices;
    uint32_t numMeshlets;
} brmi_per_mesh;                            /* 64 B */

typedef struct brmi_per_mesh_instance {
    uint32_t perMeshBufferIndex;
    uint32_t perObjectBufferIndex;
    uint32_t skinningInstanceSlot;
    float    skinnedBoundsScale;
    float    boundingSphere[4];
} brmi_per_mesh_instance;                   /* 32 B */

/* ---- cameras ----------------------------------------------------------------------------- */
/* BR/shaders/Include/structs.hlsli:148-176 */
typedef struct brmi_camera {
    float    positionWorldSpace[4];
    float    view[4][4];
    float    viewInverse[4][4];
    float    projection[4][4];
    float    projectionInverse[4][4];
    float    viewProjection[4][4];
    float    prevView[4][4];
    float    prevJitteredProjection[4][4];
    float    prevUnjitteredProjection[4][4];
    float    unjitteredProjection[4][4];
    float    clippingPlanes[6][4];   /* view-space, normalised: near, far, left, right, bottom, top
                                        (BR/src/Utilities/Utilities.cpp:1840-1868) */
    float    fov, aspectRatio, zNear, zFar;
    int32_t  depthBufferArrayIndex;
    uint32_t depthResX, depthResY, numDepthMips;
    uint32_t isOrtho;
    float    UVScaleToNextPowerOf2[2];
    uint32_t pad[1];
} brmi_camera;                              /* 736 B */

/* BR/shaders/Include/structs.hlsli:178-194 */
typedef struct brmi_culling_camera {
    float    positionWorldSpace[4];
    float    projX, projY, zNear, errorOverDistanceThreshold;
    uint32_t isOrtho;
    float    pad[3];
    float    viewRightWorld[4];
    float    viewUpWorld[4];
    float    viewForwardWorld[4];
    float    viewProjection[4][4];
    float    viewZ[4];
    float    viewInverse[4][4];
    float    projectionInverse[4][4];
} brmi_culling_camera;                      /* 304 B */

/* BR/shaders/Include/structs.hlsli:47-61 */
typedef struct brmi_view_raster_info {
    uint32_t visibilityUAVDescriptorIndex;
    uint32_t opaqueVisibilitySRVDescriptorIndex;
    uint32_t deepVisibilityHeadPointerUAVDescriptorIndex;
    uint32_t scissorMinX, scissorMinY, scissorMaxX, scissorMaxY;
    float    viewportScaleX, viewportScaleY;
    uint32_t pad0, pad1, pad2;
} brmi_view_raster_info;                    /* 48 B */

/* BR/shaders/Include/structs.hlsli:196-224 */
typedef struct brmi_per_frame {
    float    ambientLighting[4];
    float    shadowCascadeSplits[4];
    uint32_t mainCameraIndex;
    uint32_t numLights;
    uint32_t numDirectionalClipmaps;
    uint32_t activeEnvironmentIndex;
    uint32_t outputType;
    uint32_t screenResX, screenResY;
    uint32_t lightClusterGridSizeX, lightClusterGridSizeY, lightClusterGridSizeZ;
    uint32_t nearClusterCount;
    float    clusterZSplitDepth;
    uint32_t frameIndex;
    uint32_t shadowVirtualSmrtDirectionalCountsPacked;
    float    shadowVirtualSmrtMaxRayAngleFromLightDegrees;
    float    shadowVirtualSmrtRayLengthScaleDirectional;
    float    shadowVirtualSmrtMaxTraceDistanceWorld;
    float    _padSmrt;
} brmi_per_frame;                           /* 104 B */

/* ---- lights ------------------------------------------------------------------------------ */
/* BR/shaders/Include/structs.hlsli:230-250 ; BR/include/ShaderBuffers.h:377-403 */
typedef struct brmi_light_info {
    uint32_t type;
    float    innerConeAngle;      /* cos(inner) */
    float    outerConeAngle;      /* cos(outer) */
    int32_t  shadowViewInfoIndex;
    float    posWorldSpace[4];
    float    dirWorldSpace[4];
    float    attenuation[4];      /* constant, linear, quadratic (normalised), w unused */
    float    color[4];            /* rgb normalised colour, w = intensity */
    float    nearPlane, farPlane;
    int32_t  shadowMapIndex, shadowSamplerIndex;
    uint32_t shadowCaster;
    float    boundingSphere[4];
    float    maxRange;
    float    shadowSourceRadius;
    float    shadowSourceAngleDegrees;
} brmi_light_info;                          /* 128 B */

/* BR/shaders/Include/structs.hlsli:519-531 */
typedef struct brmi_light_page {
    uint32_t ptrNextPage;
    uint32_t numLightsInPage;
    uint32_t lightIndices[BRMI_LIGHTS_PER_PAGE];
} brmi_light_page;                          /* 56 B */

typedef struct brmi_light_cluster {
    float    minPoint[4];
    float    maxPoint[4];
    uint32_t numLights;
    uint32_t ptrFirstPage;
    uint32_t pad[2];
} brmi_light_cluster;                       /* 48 B */

/* ---- materials --------------------------------------------------------------------------- */
/* BR/shaders/Include/structs.hlsli:252-323 */
typedef struct brmi_material_info {
    uint32_t materialFlags;
    uint32_t baseColorTextureIndex, baseColorSamplerIndex, normalTextureIndex;
    uint32_t normalSamplerIndex, metallicTextureIndex, metallicSamplerIndex, roughnessTextureIndex;
    uint32_t roughnessSamplerIndex, emissiveTextureIndex, emissiveSamplerIndex, aoMapIndex;
    uint32_t aoSamplerIndex, heightMapIndex, heightSamplerIndex, opacityTextureIndex;
    uint32_t opacitySamplerIndex;
    float    metallicFactor, roughnessFactor, ambientStrength;
    float    specularStrength, textureScale, heightMapScale, alphaCutoff;
    float    geometricDisplacementMin, geometricDisplacementMax;
    uint32_t geometricDisplacementEnabled, perMaterialPad0;
    float    baseColorFactor[4];
    float    emissiveFactor[4];
    uint32_t baseColorChannels[4];
    uint32_t normalChannels[3];
    uint32_t compileFlagsID;
    uint32_t aoChannel, heightChannel, metallicChannel, roughnessChannel;
    uint32_t emissiveChannels[3];
    uint32_t rasterBucketIndex;
    uint32_t baseColorUvSetIndex, normalUvSetIndex, metallicUvSetIndex, roughnessUvSetIndex;
    uint32_t emissiveUvSetIndex, aoUvSetIndex, heightUvSetIndex, opacityUvSetIndex;
    uint32_t openPBRMaterialDataIndex;
    uint32_t baseColorStreamingTextureID, normalStreamingTextureID, metallicStreamingTextureID;
    uint32_t roughnessStreamingTextureID, emissiveStreamingTextureID, aoStreamingTextureID;
    uint32_t heightStreamingTextureID, opacityStreamingTextureID;
} brmi_material_info;                       /* 276 B */

/* BR/shaders/Include/structs.hlsli:383-470 (texture-binding tail kept for layout fidelity) */
typedef struct brmi_openpbr_material_info {
    float    baseWeight;
    float    baseColor[3];
    float    baseDiffuseRoughness, baseMetalness, subsurfaceWeight, subsurfaceRadius;
    float    subsurfaceColor[3];
    float    subsurfaceScatterAnisotropy;
    float    subsurfaceRadiusScale[3];
    float    specularWeight;
    float    specularColor[3];
    float    specularRoughness, specularRoughnessAnisotropy, specularIor;
    float    specularAnisotropyRotationCosSin[2];
    float    coatWeight;
    float    coatColor[3];
    float    coatRoughness, coatRoughnessAnisotropy, coatIor, coatDarkening;
    float    coatAnisotropyRotationCosSin[2];
    float    fuzzWeight;
    float    fuzzColor[3];
    float    fuzzRoughness, transmissionWeight;
    float    transmissionColor[3];
    float    transmissionDepth;
    float    transmissionScatter[3];
    float    transmissionScatterAnisotropy, transmissionDispersionScale, transmissionDispersionAbbeNumber;
    float    thinFilmWeight, thinFilmThickness, thinFilmIor, emissionLuminance;
    float    emissionColor[3];
    float    geometryOpacity;
    uint32_t geometryThinWalled, pad0, pad1, pad2;
    uint32_t textureBindings[38];   /* [0..11] coat colour / weight / roughness, fuzz colour / weight / roughness: (texture, sampler) index pairs,
                                       0xFFFFFFFF = none (OPENPBR_INVALID_TEXTURE_INDEX, utilities.hlsli:641); [12..18] coat colour channels x4, weight
                                       channel, roughness channel, pad; [19..25] the same for fuzz; [26..31] UV set per slot; [32..37] streaming ids */
} brmi_openpbr_material_info;               /* 400 B */

/* ---- textures and samplers ---------------------------------------------------------------
 * What a bindless descriptor-heap slot stands for on this boundary: `MaterialInfo::*TextureIndex` indexes
 * brmi_scene_buffers::textures, `*SamplerIndex` indexes brmi_scene_buffers::samplers.  Texels are RGBA8, row-major,
 * the mip chain tightly packed (level l starts mipOffset[l] texels after `texels`, size max(1, w >> l) x max(1, h >> l)).
 * The sampler is evaluated in software (DESIGN.md "software sampler"): rhi::SamplerDesc fields of
 * BR/src/Resources/Sampler.cpp:20-40 / BR/src/Import/GlTFLoader.cpp:856-885 that an isotropic filter reads. */
#define BRMI_TEXTURE_FORMAT_RGBA8_UNORM       0u
#define BRMI_TEXTURE_FORMAT_RGBA8_UNORM_SRGB  1u     /* rgb decoded through brmi_scene_buffers::srgbToLinear before filtering */
#define BRMI_TEXTURE_MAX_MIPS                 16u
typedef struct brmi_texture_desc {
    const uint8_t* texels;                  /* device pointer, 4-byte aligned (scene generator output: byte offset into BRMI_ARR_TEXELS) */
    uint32_t width, height, mipCount, format;
    uint32_t mipOffset[BRMI_TEXTURE_MAX_MIPS];
    uint32_t reserved[2];
} brmi_texture_desc;                        /* 96 B */
#define BRMI_ADDRESS_WRAP    0u
#define BRMI_ADDRESS_MIRROR  1u
#define BRMI_ADDRESS_CLAMP   2u
#define BRMI_FILTER_POINT    0u
#define BRMI_FILTER_LINEAR   1u
typedef struct brmi_sampler_desc {
    uint32_t addressU, addressV;
    uint32_t minFilter, magFilter, mipFilter;
    float    mipLodBias, minLod, maxLod;
} brmi_sampler_desc;                        /* 32 B */

/* ---- path-internal records --------------------------------------------------------------- */
/* 16-byte packed visible cluster, BR/shaders/Include/visibleClusterPacking.hlsli:83-122,220-235
 *   x: view:8 | instance:24      y: localMeshlet:14 | group[17:0]:18
 *   z: group[19:18]:2 | slabDescriptor:20 | pageIndex:10       w: vsm / voxel payload            */
typedef struct brmi_visible_cluster { uint32_t x, y, z, w; } brmi_visible_cluster;

/* BR/include/Render/GraphExtensions/ClusterLOD/CLodCommon.h: TraverseNodeRecord / MeshletBucketRecord */
typedef struct brmi_traverse_node_record { uint32_t instanceIndex, nodeIdPacked, viewId; } brmi_traverse_node_record; /* 12 B */
typedef struct brmi_meshlet_bucket_record {
    uint32_t instanceIndex, viewId, groupIdPacked, meshletIndexAndCount, pageSlabDescriptorIndex, pageSlabByteOffset;
} brmi_meshlet_bucket_record;               /* 24 B */

#ifdef __cplusplus
}
#endif
#endif /* BRMI_TYPES_H */
