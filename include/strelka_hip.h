/*
 * strelka_hip.h -- C ABI of the MI355X-native wavefront path tracer that sits behind
 * Strelka's oka::Render interface.
 *
 * This header is the drop-in boundary.  Every entry point names the reference interface it
 * replaces (paths relative to the arhix52/Strelka tree).  The ABI is plain C: opaque handle,
 * POD structs, host pointers + counts, int status (0 = ok).  No STL, no glm, no torch types.
 *
 * Ownership: the library owns all device memory.  The caller owns every host pointer it passes;
 * pointers only need to live for the duration of the call.  Device pointers passed in
 * (skh_render_subframe's d_image, skh_copy_accum's d_dst) are caller-owned HBM buffers.
 *
 * Threading: one context per GPU, externally synchronised (same contract as oka::Render, whose
 * render() is synchronous and not re-entrant: src/render/optix/OptixRender.cpp:874-1057).
 *
 * Errors: functions return skh_status; skh_last_error() gives the message.  The library never
 * aborts (the reference logs + assert(0): OptixRender.cpp:61-103).
 */
#ifndef STRELKA_HIP_H
#define STRELKA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKH_ABI_VERSION 5 /* 5 (round 6): + skh_refit_accel, skh_build_info.refit / ms_refit; options curve_merge, curve_segnode, curve_strand_major, split_pairs.  4 (round 5): + skh_get_build_info; options reinsert_rounds, reinsert_min_size; wide, tail_park, tail_lag removed.  3 (round 4): + skh_unit_probe, skh_copy_aov */

/* mirrors oka::Result (include/render/common.h:30-35) */
typedef enum skh_status
{
    SKH_OK = 0,
    SKH_FAIL = 1,
    SKH_OUT_OF_MEMORY = 2,
    SKH_INVALID_ARGUMENT = 3
} skh_status;

typedef struct skh_context skh_context;

/* oka::Scene::Vertex (include/scene/scene.h:80-89) == Vertex (OptixRenderParams.h:19-28): 32 B AoS.
 * normal/tangent: 10-10-10 packed (scene.cpp:111-117), uv: 16-16 packed over [-10,10] (RenderPass.cpp:53-67). */
typedef struct skh_vertex
{
    float pos[3];
    uint32_t tangent;
    uint32_t normal;
    uint32_t uv;
    float pad0;
    float pad1;
} skh_vertex;

/* oka::Mesh (include/scene/scene.h:21-27).  Indices are mesh-local; vertex_offset is added at fetch time
 * (OptixRender_radiance_closest_hit.cpp:365-376). */
typedef struct skh_mesh
{
    uint32_t index_offset; /* mIndex    */
    uint32_t index_count;  /* mCount    */
    uint32_t vertex_offset; /* mVbOffset */
    uint32_t vertex_count; /* mVertexCount */
} skh_mesh;

/* oka::Curve (include/scene/scene.h:29-42); always cubic B-spline on the render side
 * (OptixRender.cpp:218-245: degree 3, segments = n-3 per strand). */
typedef struct skh_curve
{
    uint32_t vertex_counts_start;
    uint32_t vertex_counts_count;
    uint32_t points_start;
    uint32_t points_count;
    uint32_t widths_start;
    uint32_t widths_count;
} skh_curve;

/* oka::Instance::Type (include/scene/scene.h:47-52) */
enum
{
    SKH_INSTANCE_MESH = 0,
    SKH_INSTANCE_LIGHT = 1,
    SKH_INSTANCE_CURVE = 2
};

/* oka::Instance (include/scene/scene.h:44-60) with the transform already in the 3x4 row-major
 * object-to-world form the reference hands to OptiX (OptixRender.cpp:438). 64 B. */
typedef struct skh_instance
{
    float transform[12]; /* row-major 3x4, object -> world */
    uint32_t type; /* SKH_INSTANCE_* */
    uint32_t geom_id; /* mMeshId / mCurveId */
    uint32_t material_id; /* 0xffffffff -> material 0 (OptixRender.cpp:768) */
    uint32_t light_id; /* only for SKH_INSTANCE_LIGHT */
} skh_instance;

/* oka::Scene::Light == UniformLight (include/render/Lights.h:5-14): 112 B.
 * type: 0 rect, 1 disc (no sampler/pdf in the reference), 2 sphere, 3 distant. */
typedef struct skh_light
{
    float points[4][4];
    float color[4];
    float normal[4];
    int32_t type;
    float half_angle;
    float pad0;
    float pad1;
} skh_light;

/* Fixed-layout "MDL-equivalent" material argument block (replaces the MDL SDK generated argument
 * block + PTX: OptixRender.cpp:1270-1433, materialmanager.cpp:524-609).  64 B. */
enum
{
    SKH_MAT_DIFFUSE = 0, /* default.mdl::default_material(diffuse_color)  (OptixRender.cpp:1090-1097) */
    SKH_MAT_PBR = 1, /* OmniPBR: diffuse_color_constant, reflection_roughness_constant, metallic_constant */
    SKH_MAT_GLASS = 2, /* OmniGlass: glass_color, glass_ior, frosting_roughness (thin_walled = false; gltfloader.cpp:354-406) */
    SKH_MAT_HAIR = 3 /* hair sub-expression (mdlPtxCodeGen.cpp:143-155): df::chiang_hair_bsdf */
};

/* Field use per type:
 *   DIFFUSE  base_color = diffuse_color
 *   PBR      base_color = diffuse_color_constant, roughness = reflection_roughness_constant, metallic = metallic_constant,
 *            specular = specular_level
 *   GLASS    base_color = glass_color, ior = glass_ior, roughness = frosting_roughness (< 1e-3: clear glass, specular events;
 *            else a rough dielectric, Walter et al. 2007 / GGX with alpha = roughness^2, glossy events)
 *   HAIR     the arguments of df::chiang_hair_bsdf (Chiang et al. 2016): base_color = diffuse_reflection_tint,
 *            roughness = roughness_R.x, metallic = roughness_TT.x and specular = roughness_TRT.x (<= 0: derived from R as in the
 *            paper, v_TT = v_R / 4, v_TRT = 4 v_R), ior = ior, reserved[0..2] = absorption_coefficient, reserved[3] = azimuthal
 *            roughness (roughness_*.y), reserved[4] = cuticle_angle (radians), reserved[5] = diffuse_reflection_weight */
typedef struct skh_material
{
    uint32_t type; /* SKH_MAT_* */
    float base_color[3];
    float roughness;
    float metallic;
    float specular; /* specular_level, OmniPBR default 0.5 */
    float ior;
    /* Texture ids (1-based index into skh_set_textures' list, 0 = none -- MDL numbers its resources from 1, 0 being the
     * invalid texture: texture_support_cuda.h:300-304).  OmniPBR "diffuse_texture" / "normalmap_texture"
     * (gltfloader.cpp:336-350): a valid diffuse texture replaces base_color, a valid normal map perturbs state.normal. */
    uint32_t base_color_texture;
    uint32_t normal_texture;
    float reserved[6];
} skh_material;

/* One 2-D texture as the reference loads it: stbi_load(..., STBI_rgb_alpha) -> uchar4 array, sampled with
 * cudaReadModeNormalizedFloat / cudaFilterModeLinear / cudaAddressModeWrap / normalized coordinates
 * (OptixRender.cpp:1191-1264; lookup: texture_support_cuda.h:287-313).  Row 0 first, 4 bytes per texel. */
typedef struct skh_texture
{
    const uint8_t* rgba8;
    uint32_t width, height;
} skh_texture;

/* Per-launch constants == the subset of Params (OptixRenderParams.h:38-68) that render() fills
 * every call (OptixRender.cpp:936-1004). Matrices are ROW-major (OptixRender.cpp:953-954). */
typedef struct skh_frame_params
{
    float view_to_world[16];
    float clip_to_view[16];
    uint32_t subframe_index; /* Params::subframe_index      */
    uint32_t samples_this_launch; /* Params::samples_per_launch  */
    uint32_t spp_total; /* Params::maxSampleCount      */
    uint32_t max_depth; /* Params::max_depth           */
    uint32_t rect_light_sampling_method; /* 0 uniform, 1 spherical rectangle */
    float exposure[3];
    uint32_t enable_accumulation;
    uint32_t debug; /* 0 none, 1 normals, 2 diffuse AOV, 3 specular AOV */
    float shadow_ray_tmin;
    float material_ray_tmin;
} skh_frame_params;

/* One ray for skh_trace (the optixTrace call sites: OptixRender.cu:120-129, closest_hit.cu:185-197). */
typedef struct skh_ray
{
    float origin[3];
    float tmin;
    float dir[3];
    float tmax;
} skh_ray;

/* What optixGetRayTmax / optixGetInstanceIndex / optixGetPrimitiveIndex /
 * optixGetTriangleBarycentrics / optixGetCurveParameter return.  20 B. */
typedef struct skh_hit
{
    float t; /* < 0 : miss */
    uint32_t instance_id; /* 0xffffffff : miss */
    uint32_t prim_id;
    float u; /* triangle barycentric u, or curve parameter */
    float v;
} skh_hit;

enum
{
    SKH_TRACE_CLOSEST = 0, /* visibility mask 255 (OptixRender.cu:124) */
    SKH_TRACE_SHADOW = 1 /* RAY_MASK_SHADOW, terminate on first hit; hit.t = 1 if occluded, -1 if not */
};

enum
{
    /* Which builder skh_build_accel runs (all on the GPU; hit records do not depend on the choice).  The NAMES are north_star's ("LBVH / SAH refit");
     * what they select:
     *   SKH_BUILD_LBVH (0)  the context's default = option "build_quality": 1 (default) -> the same builder as SKH_BUILD_SAH below;
     *                       0 -> a plain Karras 2012 radix tree over the Morton order (fastest build, ~25 % more node visits per ray).
     *   SKH_BUILD_SAH  (1)  the quality builder, whatever the option says: Morton sort -> PLOC agglomerative clustering (Meister & Bittner 2018a) ->
     *                       rounds of parallel reinsertion (Meister & Bittner 2018b; the sum of the internal boxes' areas is the cost it lowers) ->
     *                       collapse to 4-wide quantised nodes.  "SAH-class": it minimises a surface-area cost, it is not a sweep / binned SAH.
     * REFIT: skh_refit_accel below (topology kept, leaf records and boxes recomputed: kitchen stand-in 2.2 ms against 61 ms for the build) -- the
     * reference builds its acceleration structures once, on frame 0, and ignores later edits (OptixRender.cpp:876). */
    SKH_BUILD_LBVH = 0,
    SKH_BUILD_SAH = 1
};

typedef struct skh_stats
{
    uint64_t rays_radiance; /* closest-hit rays traced since the last reset for sub-frames that were DELIVERED (see speculated_discarded) */
    uint64_t rays_shadow; /* any-hit rays, likewise */
    /* traversal counters, counter build only (skh_set_option "count_traversal"); [0] closest-hit kernel,
     * [1] shadow (any-hit) kernel */
    uint64_t nodes_visited[2]; /* 64-byte BVH nodes fetched */
    uint64_t prims_tested[2]; /* triangles */
    uint64_t segs_tested[2]; /* curve segments */
    uint64_t instances_entered[2];
    double ms_trace_closest; /* hipEvent time summed over launches since the last reset */
    double ms_trace_shadow;
    double ms_shade;
    double ms_raygen;
    double ms_accumulate;
    double ms_build; /* last skh_build_accel */
    double ms_sort; /* always 0 (ray re-ordering, a measured negative of round 1, was removed in round 3) */
    uint32_t launches_trace_closest;
    uint32_t launches_trace_shadow;
    uint32_t launches_shade;
    uint32_t launches_other;
    /* render / trace calls since the last reset that returned SKH_FAIL because a traversal stack (20 LDS + 104 global entries per
     * ray) overflowed and dropped a subtree -- a degenerate hierarchy; such a call's hits may be incomplete */
    uint32_t stack_overflows;
    /* sub-frames skh_render_subframe traced ahead (option speculate) and had to throw away since the last reset -- the caller moved
     * the camera, changed the scene or resized; their rays are NOT in rays_radiance / rays_shadow */
    uint32_t speculated_discarded;
} skh_stats;

/* ---- lifetime: RenderFactory::createRender + Render::init (render.cpp:10-35, OptixRender.cpp:1059-1105) ---- */
skh_status skh_create(int device_ordinal, skh_context** out_ctx);
void skh_destroy(skh_context* ctx);
const char* skh_last_error(const skh_context* ctx);
uint32_t skh_abi_version(void);

/* ---- scene upload: createVertexBuffer/IndexBuffer/PointsBuffer/WidthsBuffer/LightBuffer
 *      (OptixRender.cpp:1117-1189), reading oka::Scene getters (scene.h:229-327) ---- */
skh_status skh_set_geometry(skh_context* ctx, const skh_vertex* verts, uint32_t n_verts, const uint32_t* indices,
                            uint32_t n_indices, const skh_mesh* meshes, uint32_t n_meshes);
skh_status skh_set_curves(skh_context* ctx, const float* points_xyz, uint32_t n_points, const float* radii,
                          uint32_t n_radii, const uint32_t* vertex_counts, uint32_t n_vertex_counts,
                          const skh_curve* curves, uint32_t n_curves);
skh_status skh_set_instances(skh_context* ctx, const skh_instance* instances, uint32_t n_instances);
skh_status skh_set_lights(skh_context* ctx, const skh_light* lights, uint32_t n_lights);
/* Replaces the texture list (count may be 0).  Host pointers need only live for the call. */
skh_status skh_set_textures(skh_context* ctx, const skh_texture* textures, uint32_t count);
skh_status skh_set_materials(skh_context* ctx, const skh_material* materials, uint32_t n_materials);

/* ---- createAccelerationStructure (OptixRender.cpp:388-496): per-mesh / per-curve BLAS + one TLAS ---- */
skh_status skh_build_accel(skh_context* ctx, uint32_t flags);
/* After a VERTEX edit -- skh_set_geometry with the mesh table and index buffer of the last build, any vertex data; skh_set_curves with the curve sets and vertex
 * counts of the last build, any control points and radii (an animated groom) -- keep the hierarchies' topology and
 * recompute its leaf records and boxes bottom-up (north_star's "SAH refit"; the reference has no equivalent: it builds once, on frame 0, OptixRender.cpp:876).
 * Refits when every mesh instance is baked to world space (no top level: what a bake without mesh sharing gives) and nothing but the vertices changed since
 * the build; otherwise it IS skh_build_accel(flags of the last build).  skh_build_info.refit says which happened.  Hit records do not depend on the hierarchy,
 * so a refitted tree returns what a rebuilt one returns; after large deformations a rebuild traverses faster.  Kitchen stand-in (23 M world-space triangles,
 * 6.06 M nodes): build 61 ms, refit 2.2 ms. */
skh_status skh_refit_accel(skh_context* ctx);

/* Which instances skh_build_accel baked to world space (option bake_world; `flags` receives one byte per instance, 1 = baked;
 * either pointer may be NULL).  A baked mesh instance has no IAS entry (OptixRender.cpp:412-441 creates one per instance): its
 * triangles are transformed once and intersected in world space.  Hit records name the same instance and primitive. */
skh_status skh_get_baked(skh_context* ctx, uint8_t* flags, uint32_t n_instances, uint32_t* out_baked_instances,
                         uint32_t* out_baked_triangles);

/* ---- updatePathtracerParams (OptixRender.cpp:827-872): (re)allocates accum/AOV buffers, resets history ---- */
skh_status skh_resize(skh_context* ctx, uint32_t width, uint32_t height);

/* Multi-GPU pixel-tile ownership (new; the reference is single-GPU).  tile_xy holds n_tiles (x0,y0)
 * pairs of tile_size x tile_size tiles this context renders; NULL / 0 = the whole image. */
skh_status skh_set_tiles(skh_context* ctx, uint32_t tile_size, const uint32_t* tile_xy, uint32_t n_tiles);

/* ---- OptiXRender::render's optixLaunch (OptixRender.cpp:1006-1021): one sub-frame batch.
 *      d_image may be NULL (only accum is updated).  Synchronous, like the reference: the image is complete when the call returns.
 *      Called once per sub-frame the library traces ahead (options speculate, speculate_async); the image is then written on a stream
 *      of the library's own that does NOT wait for the caller's null-stream work: d_image must have no writes of the caller's still
 *      pending when the call is made (the reference's callers only ever read it, RenderPass.cpp:441-447). ---- */
skh_status skh_render_subframe(skh_context* ctx, const skh_frame_params* params, void* d_image);
/* Called once per sub-frame, as HdStrelkaRenderPass::_Execute calls render() (RenderPass.cpp:441-447), the library traces ahead:
 * once two consecutive calls continue the same frame (identical parameters, subframe_index + 1) the next call traces 2, then 4,
 * ... up to option "speculate" (8) sub-frames in one wavefront pass, and the calls after it only apply their accumulation step.
 * The images are bit-identical to one-pass-per-call rendering (tests: test_speculative_subframes_are_exact); a call that does
 * not continue the frame -- camera motion restarts at sub-frame 0 -- simply discards what was traced ahead.  "speculate" 0 = off. */

/* Convenience for benchmarks: n consecutive sub-frames of params->samples_this_launch samples each,
 * starting at params->subframe_index, with a single device synchronisation at the end. */
skh_status skh_render_subframes(skh_context* ctx, const skh_frame_params* params, uint32_t n_subframes,
                                void* d_image);

/* ---- the tonemap() + gammaCorrection post pass (postprocessing/Tonemappers.cu:111-135), in place on a
 *      device float4 image.  type: 0 none, 1 Reinhard, 2 ACES fitted, 3 ACES film; gamma <= 0 = off ---- */
skh_status skh_tonemap(skh_context* ctx, void* d_image, uint32_t width, uint32_t height, uint32_t type,
                       const float exposure[3], float gamma);

/* ---- device image buffers: OptixBuffer's cudaMalloc / cudaFree / cudaMemcpy (OptixBuffer.cpp:16-43, 45-63) ---- */
skh_status skh_buffer_alloc(skh_context* ctx, size_t bytes, void** out_device_ptr);
skh_status skh_buffer_free(skh_context* ctx, void* device_ptr);
skh_status skh_buffer_download(skh_context* ctx, const void* device_ptr, void* host, size_t bytes);
/* Page-locks / releases a caller-owned host range so that skh_buffer_download into it runs at PCIe rate (the host mirror of
 * Buffer::map, OptixBuffer.cpp:37-43, is a pageable std::vector in the reference; the caller maps after EVERY sub-frame). */
skh_status skh_host_register(skh_context* ctx, void* host, size_t bytes);
skh_status skh_host_unregister(skh_context* ctx, void* host);

/* ---- read-back: Buffer::map (OptixBuffer.cpp:37-43) ---- */
skh_status skh_read_accum(skh_context* ctx, float* host_rgba); /* W*H float4, row-major, row 0 = launch y 0 */
skh_status skh_read_aov(skh_context* ctx, uint32_t which /*0 diffuse, 1 specular*/, float* host_rgba);
skh_status skh_copy_accum(skh_context* ctx, void* d_dst_rgba); /* device-to-device, W*H float4 */
/* the diffuse (0) / specular (1) AOV accumulator, device-to-device: what render() copies to the image when all samples are done and
 * the debug view is 2 / 3 (OptixRender.cpp:1029-1042) */
skh_status skh_copy_aov(skh_context* ctx, uint32_t which, void* d_dst_rgba);
/* compact tile accumulators of this context: n_tiles * tile_size^2 float4, tile-major (for the RCCL gather) */
skh_status skh_copy_accum_tiles(skh_context* ctx, void* d_dst);
/* scatter gathered compact tiles into a W*H float4 image (root side of the gather) */
skh_status skh_scatter_tiles(skh_context* ctx, const void* d_src_tiles, const uint32_t* tile_xy, uint32_t n_tiles,
                             uint32_t tile_size, void* d_dst_rgba, uint32_t width, uint32_t height);

/* ---- multi-GPU: the one collective of a tile-sharded frame, below the C ABI (RCCL over xGMI; the reference is single-GPU,
 *      include/render/render.h:19-56 has no counterpart).  One communicator per context, bootstrapped the NCCL way: rank 0
 *      calls skh_comm_unique_id, the host distributes the 128 bytes (pipe, file, MPI, torch.distributed ...), every rank
 *      calls skh_comm_init (collective).  skh_gather_tiles: every rank's compact tile accumulators (the layout of
 *      skh_copy_accum_tiles, zero-padded to max_tiles tiles) -> the root's d_recv = [world][max_tiles][tile_size^2] float4, as
 *      ONE group of point-to-point sends into the root on the context's stream -- each sender uses its own xGMI link, no ring.
 *      Synchronous like render().  world_size 1 needs no communicator (a device-to-device copy).  librccl is loaded on first
 *      use; a box without it gets SKH_FAIL from skh_comm_*, nothing else depends on it. ---- */
#define SKH_COMM_ID_BYTES 128
skh_status skh_comm_unique_id(void* out_id /* SKH_COMM_ID_BYTES */);
skh_status skh_comm_init(skh_context* ctx, const void* id, int world_size, int rank);
skh_status skh_comm_destroy(skh_context* ctx);
/* world size / rank of the context's communicator and the rank count RCCL itself reports for it (ncclCommCount; 0 = none) */
skh_status skh_comm_info(skh_context* ctx, int* out_world_size, int* out_rank, int* out_rccl_ranks);
skh_status skh_gather_tiles(skh_context* ctx, uint32_t max_tiles, void* d_recv /* root only, else NULL */, int root);

/* ---- ray queries against the built accel (the optixTrace call sites), for tests and micro-benchmarks.
 *      rays/hits are HOST arrays. ---- */
skh_status skh_trace(skh_context* ctx, const skh_ray* rays, uint32_t n_rays, uint32_t mode, skh_hit* hits);
/* same on caller-owned DEVICE arrays (skh_ray / skh_hit records), the traversal repeated `repeat` times (micro-benchmarks);
 * returns once the hits are in d_hits */
skh_status skh_trace_device(skh_context* ctx, const void* d_rays, uint32_t n_rays, uint32_t mode, void* d_hits,
                            uint32_t repeat);

/* ---- memory ceilings of this GPU, measured with the access shapes of the hot path ("report a measured STREAM-copy
 *      ceiling"; the reference has no counterpart).  bench.py puts them beside the roofline fractions.  `bytes` = size of the
 *      probed buffer (allocated and freed inside; COPY takes two).  COPY: uint4 grid-stride copy, rate counts read + write.
 *      GATHER / CHASE: one-wave workgroups on the trace kernels' grid, every lane fetches 256 pseudo-random aligned records of
 *      record_bytes (32, 64 = one BVH node, or 128 = one cache line) -- four independent fetches in flight, or each address taken
 *      from the record before it (a traversal step); rate = record_bytes x fetches / time, whatever cache level served them. ---- */
#define SKH_PROBE_COPY 0u
#define SKH_PROBE_GATHER 1u
#define SKH_PROBE_CHASE 2u
skh_status skh_probe_memory(skh_context* ctx, uint32_t kind, uint64_t bytes, uint32_t record_bytes, uint32_t repeat, double* out_gbps,
                            double* out_ms);

/* ---- BSDF probes for tests: mdlcode_sample and mdlcode_evaluate (the call protocol of closest_hit.cu:563-605) run on the
 *      device for n independent inputs against the context's material list; host arrays.  One query = one MDL state
 *      (normal, geom_normal, tangent_u[0]) + k1 + the four xi of sample() + the k2 handed to evaluate(). ---- */
typedef struct skh_bsdf_query
{
    float normal[3], geom_normal[3], tangent_u[3];
    float k1[3]; /* outgoing direction (= -ray direction) */
    float k2[3]; /* incoming direction for evaluate() */
    float xi[4];
    uint32_t material; /* index into the list set by skh_set_materials */
    uint32_t inside; /* selects ior1 / ior2 as closest_hit.cu:496-498 */
} skh_bsdf_query;
typedef struct skh_bsdf_result
{
    float k2[3], bsdf_over_pdf[3], pdf; /* Bsdf_sample_data */
    int32_t event_type; /* mi::neuraylib::Bsdf_event_type bits */
    float bsdf_diffuse[3], bsdf_glossy[3], eval_pdf; /* Bsdf_evaluate_data: both include the cosine */
    uint32_t reserved0;
} skh_bsdf_result;
skh_status skh_bsdf_probe(skh_context* ctx, const skh_bsdf_query* queries, uint32_t n, skh_bsdf_result* results);

/* ---- unit probes (tests) ----
 * The device functions of the sampler (A2), the light samplers / pdfs / MIS weight (A4, A5) and the accumulator (A10), run on the GPU
 * one call per record, so that GPU tests can hold the HIP code against the fixtures the REFERENCE's own headers produced
 * (tests/golden/, generated by oracle/ref_golden.cpp).  Reference code probed: RandomSampler.h:130-137,166-175,213-226;
 * Lights.h:28-84,201-362; OptixRender.cu:60-78 + postprocessing/Utils.h:5-14.  Records are packed 32-bit words:
 *   unit                    consts               in (per record)                          param                         out (per record)
 *   SKH_UNIT_SAMPLER        --                   u32 x, y, sampleIndex, depth, dim        sppTotal                      u32 bits of random<dim>, bits of the same
 *                                                                                                                       value through the LDS-table path k_shade uses, sampleIdx
 *   SKH_UNIT_SOBOL          --                   u32 index, dim                           --                            u32 sobol_uint
 *   SKH_UNIT_LIGHT_SAMPLE   UniformLight (112 B) f32 P[3], u[2]                           0 rect uniform | 1 spherical  f32 pointOnLight[3], pdf, normal[3], area,
 *                                                                                         rect | 2 sphere | 3 distant   L[3], distToLight
 *   SKH_UNIT_LIGHT_PDF      UniformLight         f32 lightHitPoint[3], surfacePoint[3]    --                            f32 getLightPdf
 *   SKH_UNIT_LIGHT_NORMAL   UniformLight         f32 hitPoint[3]                          --                            f32 calcLightNormal[3], calcLightArea
 *   SKH_UNIT_MIS            --                   f32 a, b                                 --                            f32 misWeightBalance(a, b)
 *   SKH_UNIT_ACCUMULATE     f32 exposure[3]      f32 value[3]  (a SEQUENCE: record k is   first sub-frame index         f32 accumulator[3] after record k
 *                                                folded into the result of 0..k-1)
 *   SKH_UNIT_TONEMAP        f32 exposure[3]      f32 color[3]                             --                            f32 tonemap[3], inverseTonemap[3]
 *   SKH_UNIT_LIBM           --                   f32 x, y                                 --                            f32 sin x, cos x, acos x, asin x, atan2(y, x), exp x, log x,
 *                                                                                                                       sinh x, pow(x, y), atan2(x, y): strelka_amd/csrc/skh_libm.h,
 *                                                                                                                       the libm stand-ins both sides compile */
typedef enum skh_unit
{
    SKH_UNIT_SAMPLER = 0,
    SKH_UNIT_SOBOL = 1,
    SKH_UNIT_LIGHT_SAMPLE = 2,
    SKH_UNIT_LIGHT_PDF = 3,
    SKH_UNIT_LIGHT_NORMAL = 4,
    SKH_UNIT_MIS = 5,
    SKH_UNIT_ACCUMULATE = 6,
    SKH_UNIT_TONEMAP = 7,
    SKH_UNIT_LIBM = 8,
    SKH_UNIT_COUNT = 9
} skh_unit;
skh_status skh_unit_probe(skh_context* ctx, uint32_t unit, uint32_t param, const void* consts, const void* in, uint32_t n, void* out);

/* ---- options / stats ----
 * None of the options changes a result: hit records and images are bit-identical for every setting
 * (tests/test_gpu_parity.py::test_results_do_not_depend_on_the_acceleration_structure_or_scheduling) -- with ONE exception,
 * bake_world, which is part of the intersection's definition: a baked instance's triangles are tested in world space, so the
 * t, u, v of hits on it differ in the last bits from the object-space test (same instance, same primitive; the CPU oracle
 * takes the same setting and the bit-exact contract holds per setting).
 *   measurement   count_traversal 0|1 (counter build of the trace kernels), timing 0|1 (per-kernel hipEvent spans)
 *   scheduling    waves_per_cu (28) / waves_per_cu_shadow (28) (7 waves per SIMD) / waves_per_cu_world, waves_per_cu_shadow_world (32 / 32: the world-only builds run 8), fetch_min_closest / fetch_min_shadow (32 / 48, scenes with curves 16 / 16: idle lanes before a wave refills), fetch_min_closest_small (48: the closest-hit threshold of small overlapped passes of triangle scenes, unless fetch_min_closest is set),
 *                 merge_light_proxies (0; 1: baked light proxies share the baked mesh triangles' world-space tree and any-hit queries skip their triangles -- closest-hit -2 ... -4 %, any-hit +8 ... +13 %: measured a net loss; 0: a tree of their own, visited by the radiance rays that meet the box around all of them),
 *                 compact_hits (1: render passes of scenes the world-only kernels trace, whose instance count and primitive range share 32 bits, keep 16-byte hit records {t, u, v, instance | record or segment}; 0: 32-byte records always),
 *                 direct_records (-1 = by the counts: a baked mesh triangle's hit names its shading record, unless only (instance, mesh-local primitive) fits a 16-byte hit record; 0 / 1 forced),
 *                 fetch_chunk (-1 = auto: 128 when the triangle + curve trees hold <= 16384 nodes, else 0; queue positions a trace wave reserves per atomic on its shard's cursor),
 *                 node_break_closest / node_break_shadow (32 / 28, curves 20 / 20 until one is set explicitly: leave the node loop below x/64 descending rays),
 *                 leaf_min (16: lanes for the minority kind of leaf work), curve_min (48: lanes parked in front of the
 *                 curve intersector before it runs), subframe_batch (0 = auto: ~64 M paths per pass),
 *                 speculate (8: sub-frames traced ahead when skh_render_subframe is called once per sub-frame; 0 = off),
 *                 speculate_async 1|0 (the pass after the one being collected is traced meanwhile, on the render stream, into a second
 *                 path-state buffer; accumulation steps and skh_buffer_download run beside it on their own stream: the caller's map()
 *                 copies cost nothing any more.  A camera move waits for the pass in flight, <= `speculate` sub-frames),
 *                 overlap 0|1|2 (any-hit launches on a second stream beside the next closest-hit launch: off | small passes |
 *                 always), small_waves_closest / small_waves_shadow (shares of a 32-wave CU the two overlapped launches of a small pass take, scaled to what their build fits; 0 = automatic: 20 / 12 for triangle scenes -- the closest-hit launch is the one on the critical path --, 16 / 16 with curves; >= 32 = the full grid), small_waves_first / small_waves_last (the same for the pass's FIRST closest-hit and LAST any-hit launch, which run alone; 0 = automatic: as the others, the last any-hit launch of a triangle scene 20),
 *                 tail_split (1 | 2 | 0 | -1: the world-only triangle kernels' build whose waves, once the ray queue is dry, hand stack entries of their last rays to their idle lanes -- a long ray's
 *                 subtrees walked side by side, results merged by the closest-hit rule: same records --; 1 = every launch of a scene whose hierarchy has more than 16 384 nodes, the default; 2 = every launch; -1 = passes of
 *                 2^17 ... 2^23 paths only; 0 = never)
 *   definition    bake_world 4|3|2|1|0 (mesh instances intersected in world space, no instance entry: 1 = instances whose mesh has one
 *                 user -- what HdStrelka's per-instance meshes are --, 2 = also instances of meshes with <= bake_small_tris (64)
 *                 triangles when that empties the top level, 3 = every mesh instance, 4 (default) = 3 while the instanced triangles stay
 *                 within bake_budget_mtris (64) million, else 2; 0 = every instance keeps its TLAS leaf), world_kernel 1|0 (scenes
 *                 with an empty top level run the world-only build of the traversal kernel; curve instances -- at most 16, no unbaked
 *                 mesh or light instance beside them -- do not need a top level either: their trees are walked from the world-only kernel with the
 *                 curve block, each instance's transform applied to the ray as at a TLAS leaf)
 *   build         build_quality 1|0 (PLOC | Karras radix tree), reinsert_rounds (8; 0 = off: rounds of parallel reinsertion over the PLOC tree of the
 *                 triangle build -- every subtree looks for the place in the tree where it costs least, the best non-conflicting moves are applied),
 *                 reinsert_curve_rounds (4: the same over the curve sub-segment trees),
 *                 reinsert_min_size (0 = auto: 1 up to 4 M triangles, 32 beyond: only the part of the tree above subtrees of this many primitives is optimised), morton_bits (10 per axis in the sort keys; 4..21), ploc_top (0: clusters left at which the triangle build widens PLOC's neighbour search from 12 to 96), leaf_max_tris (2), leaf_lines 0|1 (triangle leaves padded so that none
 *                 straddles a 128-byte line it need not: -11 % fetched lines, same time, more memory), curve_leaf (1), curve_split (4: parameter sub-ranges
 *                 per curve segment), tlas_build 1|0|2 (GPU PLOC over the instance boxes (default) | exact sweep SAH on the host: 5 % fewer instance
 *                 entries, single-threaded | the sweep up to 8192 instances, the GPU beyond), tlas_open (1: TLAS leaves per instance budget),
 *                 tight_instance_boxes 1|0
 *   (round 4's `wide` 8, `tail_park`, `tail_lag` are gone: measured negatives, experiments/README.md)
 * Unknown names and out-of-range values return SKH_INVALID_ARGUMENT. */
skh_status skh_set_option(skh_context* ctx, const char* name, int64_t value);
/* what the context's device reports (hipDeviceProp_t): the measurement code prices instruction rates against these */
typedef struct skh_device_info
{
    uint32_t compute_units; /* 256 on MI355X */
    uint32_t simds_per_cu; /* 4 */
    uint32_t clock_khz; /* maximum engine clock */
    uint32_t memory_clock_khz;
    uint32_t memory_bus_bits;
    uint32_t wavefront_size;
    uint64_t total_memory_bytes;
    char name[64];
} skh_device_info;
skh_status skh_get_device_info(skh_context* ctx, skh_device_info* out);
/* what the last skh_build_accel did to the triangle hierarchy (the builder that replaces optixAccelBuild with PREFER_FAST_TRACE,
 * OptixRender.cpp:318-386): PLOC over the Morton order, then rounds of parallel reinsertion (options reinsert_rounds / reinsert_min_size) */
typedef struct skh_build_info
{
    uint32_t triangles; /* primitives of the triangle build (meshes + baked world-space groups) */
    uint32_t nodes; /* 64-byte 4-wide nodes */
    uint32_t reinsert_rounds; /* rounds that ran (a round without a move ends the pass) */
    uint32_t reinsert_moves; /* subtrees moved, all rounds */
    uint32_t reinsert_min_size; /* truncation used: only nodes whose parent holds at least this many primitives moved */
    uint32_t refit; /* 1: the hierarchy in use came out of skh_refit_accel's refit (topology of the last build, boxes of the current vertices); 0: out of a build */
    double cost_before; /* sum of the internal binary nodes' box half-areas after PLOC ... */
    double cost_after; /* ... and after the reinsertion pass (== cost_before when it did not run) */
    double ms_reinsert; /* wall time of the pass, inside ms_build */
    double ms_build; /* == skh_stats.ms_build */
    double ms_refit; /* wall time of the last refit (leaf records gathered again + one launch per tree level + the shading tables) */
} skh_build_info;
skh_status skh_get_build_info(skh_context* ctx, skh_build_info* out);
skh_status skh_get_stats(skh_context* ctx, skh_stats* out);
skh_status skh_reset_stats(skh_context* ctx);
skh_status skh_synchronize(skh_context* ctx);
void* skh_get_stream(skh_context* ctx); /* hipStream_t the kernels are launched on */

#ifdef __cplusplus
}
#endif
#endif /* STRELKA_HIP_H */
