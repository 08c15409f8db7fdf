/*
 * ray_tracing_amd — C ABI of the MI355X-native path tracer (libray_tracing_amd.so).
 *
 * The reference (damnkk/Ray_tracing) has no plugin/FFI API: its operator boundary is the OpenGL hand-off in
 * path_tracing_5chap/path_tracing_5chap.cpp (abbrev. c5) — two RGB32F texture buffers, two RGB32F 2-D textures, a few
 * uniforms, one full-screen draw per frame.  Each entry point below cites the reference interface it replaces
 * (c2 = path_tracing_2chap/path_tracing_2.cpp, c4 = path_tracing_4chap/path_tracing_4.cpp,
 *  f3/f4/f5 = path_tracing_{3,4,5}chap/shaders/fshader.fsh, hdr = path_tracing_5chap/lib/hdrloader.cpp).
 *
 * Conventions: every function returns RT_OK (0) or a negative rt_status; rt_last_error() gives the message of the last
 * failure on the calling thread.  Nothing aborts.  All floats are IEEE fp32; all arrays are plain C arrays.
 * Images are in GL row order (row 0 = bottom row), 3 floats per pixel, as pass1 writes them (f5:987).
 */
#ifndef RAY_TRACING_AMD_H
#define RAY_TRACING_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = -1,
    RT_ERR_IO = -2,
    RT_ERR_FORMAT = -3,
    RT_ERR_HIP = -4,
    RT_ERR_NO_DEVICE = -5,
    RT_ERR_BAD_BVH = -6,
    RT_ERR_OOM = -7
} rt_status;

const char* rt_last_error(void);

/* ------------------------------------------------------------------------------------------------------------------
 * Host scene layer (CPU, C++): produces exactly the arrays the reference uploads.
 * ------------------------------------------------------------------------------------------------------------------ */

/* Disney material parameter block — field order of `struct Material`, c5:23-39 (= texels 6..11 of Triangle_encoded). */
typedef struct rt_material {
    float emissive[3];
    float baseColor[3];
    float subsurface, metallic, specular;
    float specularTint, roughness, anisotropic;
    float sheen, sheenTint, clearcoat;
    float clearcoatGloss, IOR, transmission;
} rt_material;

/* Defaults of c5:25-38: emissive 0, baseColor 1, IOR 1, everything else 0. */
void rt_material_default(rt_material* m);

/* A growable list of Triangle_encoded records (c5:54-64): 36 floats = 144 B per triangle,
 * p1 p2 p3 n1 n2 n3 emissive baseColor param1..param4. */
typedef struct rt_mesh rt_mesh;
int rt_mesh_create(rt_mesh** out);
void rt_mesh_destroy(rt_mesh* m);
int rt_mesh_count(const rt_mesh* m);
const float* rt_mesh_data(const rt_mesh* m); /* valid until the mesh is next modified */
int rt_mesh_append(rt_mesh* m, const float* tris36, int n_triangles);

/* getTransformMatrix, c5:249-265: translate * rotX*rotY*rotZ * scale, angles in degrees; column-major 4x4 out. */
int rt_transform_matrix(const float rotate_deg[3], const float translate[3], const float scale[3], float out16[16]);

/* readObj of chapters 3-5, c5:267-397: `v`, `f a b c | a/b | a/b/c` (by slash count), unit max-extent normalisation,
 * `trans` (column-major), smooth (area-unweighted) or flat normals, `material` copied per triangle; appends. */
int rt_mesh_read_obj(rt_mesh* m, const char* path, const rt_material* material, const float trans16[16], int smooth_normal);

/* readObj of chapter 2, c2:142-183 (positions + `f a b c` only, no normalisation), followed by the vertex edit of
 * c2:590-594 (`v *= scale; v.y -= y_offset`).  Triangles get flat normals and the default material; appends. */
int rt_mesh_read_obj_chap2(rt_mesh* m, const char* path, float scale, float y_offset);

/* buildBVHwithSAH (use_sah != 0, c5:462-593 / c2:309-436) or buildBVH (c5:399-460) over the whole mesh with leaf size
 * `leaf_n` (the reference passes 8, c5:843).  Permutes the mesh in place exactly as the reference's std::sort calls do
 * and returns BVHNode_encoded records (c5:65-70; 12 floats = 48 B per node) with the dummy node at index 0 and the
 * root at index 1 (c5:835-843).  Node 0's fields are zeroed (the reference leaves `index` uninitialised).
 * `*nodes_out` is released with rt_free. */
int rt_bvh_build(rt_mesh* m, int leaf_n, int use_sah, float** nodes_out, int* n_nodes);

/* HDRLoader::load, hdr:29-97 (Radiance RGBE, new and old RLE; rows in file order, top scanline first).
 * The resolution line is parsed with %d (the reference's %ld into int is UB on LP64).  `*rgb` released with rt_free. */
int rt_hdr_load(const char* path, int* width, int* height, float** rgb);

/* calculateHdrCache, c5:595-698: W*H*3 floats (x/W, y/H, pdf) from marginal/conditional CDF inversion. */
int rt_hdr_cache(const float* rgb, int width, int height, float* cache_out);

/* The same table computed on the GPU (current HIP device), bit-identical to rt_hdr_cache: calculateHdrCache, c5:595-698 — the
 * fp32 running sum lumSum in the reference's row-major order (c5:602-611) by one wave, the marginal / conditional prefix sums
 * (c5:622-651) by one lane per column, the per-texel std::lower_bound searches (c5:665-682) in parallel.  SURVEY 8(f)4; the BVH
 * build stays on the host ("Start-up").  Host pointers in and out; RT_ERR_NO_DEVICE without a GPU. */
int rt_hdr_cache_gpu(const float* rgb, int width, int height, float* cache_out);

/* Orbit camera of display(), c5:716-720: eye = r*(-sin(rot)cos(up), sin(up), cos(rot)cos(up)),
 * cameraRotate = inverse(lookAt(eye, 0, +Y)), column-major. */
int rt_camera(float rotat_angle_deg, float up_angle_deg, float r, float eye[3], float camera_rotate16[16]);

/* pass3.fsh:14-25: c/(1+lum/1.5), pow(c,1/2.2), then GL's clamp + round-to-nearest 8-bit conversion.
 * flip_y != 0 writes the top row first (what a PNG wants). */
int rt_tonemap_rgb8(const float* rgb, int width, int height, unsigned char* out_rgb8, int flip_y);

/* svpng(), path_tracing_1chap/svpng.inc:77-108: stored-deflate PNG, byte-identical output.  alpha != 0 => RGBA in. */
int rt_write_png(const char* path, int width, int height, const unsigned char* img, int alpha);

void rt_free(void* p);

/* On-disk scene cache (no counterpart in the reference, which rebuilds its BVH at every start, c5:843).
 * The file holds the two arrays the reference uploads — Triangle_encoded[] (c5:54-64) in its final permuted order and
 * BVHNode_encoded[] (c5:65-70) — raw little-endian fp32 behind a 48-byte header (magic "RTAMDSC2", record sizes, `key`,
 * counts, and a payload hash: FNV-1a-64 over the FNV-1a-64 hashes of the payload's 4 MiB blocks, hashed by all cores).  `key` is the caller's hash of whatever the arrays were built from (OBJ bytes,
 * materials, transforms, leaf size): rt_scene_cache_load refuses a file whose key differs (RT_ERR_FORMAT), is truncated
 * or whose payload hash is wrong.  Loaded arrays are released with rt_free.  rt_hash_fnv1a64(seed = 0) starts a hash;
 * pass the previous result as `seed` to chain buffers. */
uint64_t rt_hash_fnv1a64(const void* data, uint64_t n_bytes, uint64_t seed);
int rt_scene_cache_save(const char* path, uint64_t key, const float* tris, int n_triangles, const float* nodes, int n_nodes);
int rt_scene_cache_load(const char* path, uint64_t key, float** tris_out, int* n_triangles, float** nodes_out, int* n_nodes);

/* Chapter 1 (config 1, host only): the CPU Cornell-box tracer of path_tracing_1chap/path_tracing.cpp — scene literal
 * c1:355-400, shoot c1:214-225, pathTracing c1:277-324, pixel loop c1:409-475 — race-free (one mt19937 per image row,
 * seeded from `seed`; the reference shares one generator between 50 threads).  `image` = W*H*3 doubles, row 0 = top
 * (c1:404, 417); rt_chap1_to_rgb8 = imshow(), c1:326-347 (gamma 1/1.9, truncation). */
int rt_chap1_render(int width, int height, int samples, unsigned seed, double* image);
int rt_chap1_to_rgb8(const double* image, int width, int height, unsigned char* out_rgb8);

/* ------------------------------------------------------------------------------------------------------------------
 * Device layer (HIP, gfx950): the per-pixel path loop.
 * ------------------------------------------------------------------------------------------------------------------ */

/* Replaces the GL context: binds the calling thread's library state to HIP device `device` and makes it current.
 * Scenes created afterwards on this thread live on that device.  Without rt_init, rt_scene_create adopts the device
 * that is current on the calling thread (hipGetDevice).  Every other entry point makes its scene's device current for
 * the duration of the call and restores the caller's current device before it returns. */
int rt_init(int device);

/* Replaces the TBO / texture uploads of c5:881-909.  `tris` = Triangle_encoded[nT] (36 floats each), `nodes` =
 * BVHNode_encoded[nN] (12 floats each, root at 1), `hdr_map`/`hdr_cache` = RGB32F W*H (row 0 = first uploaded row) or
 * NULL.  The library copies (like glBufferData); the BVH is validated (RT_ERR_BAD_BVH) before any kernel can see it. */
typedef struct rt_scene rt_scene;
int rt_scene_create(const float* tris, int n_triangles, const float* nodes, int n_nodes, const float* hdr_map,
                    const float* hdr_cache, int hdr_width, int hdr_height, rt_scene** out);
/* The same with options (NULL = defaults).  No counterpart in the reference. */
typedef struct rt_scene_options {
    int32_t entry_bvh;         /* 0: build the device-private entry BVH (rt_tuning.no_entry_bvh) where the caller's BVH is expensive
                                  to walk (cost_reference_tree > 32, >= 1024 triangles, at least twice as cheap); 1: always build
                                  it (if the BVH allows culling at all); -1: never */
} rt_scene_options;
int rt_scene_create_ex(const float* tris, int n_triangles, const float* nodes, int n_nodes, const float* hdr_map,
                       const float* hdr_cache, int hdr_width, int hdr_height, const rt_scene_options* options, rt_scene** out);
void rt_scene_destroy(rt_scene* s);

enum { RT_MODE_CHAP3 = 3, RT_MODE_CHAP4 = 4, RT_MODE_CHAP5 = 5 }; /* which fshader.fsh main() to run */
enum { RT_ACCUM_RUNNING_MEAN = 0, /* color = mix(last, color, 1/(frameCounter+1)), f5:954-955 */
       RT_ACCUM_SUM = 1 /* accum += color (spp-split across GPUs; divide after the reduce) */ };

/* The uniforms of f5:8-23 / c5:726-730, 923-928. */
typedef struct rt_render_desc {
    int32_t width, height;     /* uniform width,height; 1..65535 x 1..32767 */
    int32_t mode;              /* RT_MODE_* */
    int32_t max_bounce;        /* literal 4 at f5:949, 6 at f4:564 / f3; <=0 selects the reference literal */
    int32_t accumulate;        /* RT_ACCUM_* */
    float eye[3];              /* uniform eye */
    float camera_rotate[16];   /* uniform cameraRotate, column-major (GL_FALSE transpose, c5:728) */
} rt_render_desc;

/* pass1.draw() for frameCounter = frame_begin .. frame_begin+frame_count-1 (c5:729, 755).  `accum` (W*H*3 fp32) is
 * `lastFrame` on entry and the new `lastFrame` on return.  rt_render takes host memory and is synchronous;
 * rt_render_device takes device memory and enqueues on `hip_stream` (a hipStream_t, NULL = default stream). */
int rt_render(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, float* accum_host);
int rt_render_device(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, float* accum_dev,
                     void* hip_stream);

/* hitBVH of chapter 2 (c2:490-513) for the un-jittered camera rays of rt_render_desc: per pixel the index of the
 * closest triangle (-1 = miss) and its distance, with chapter 2's tie rule (equal distance => the candidate in the
 * right-most leaf, lowest index within a leaf). */
int rt_primary_hits(rt_scene* s, const rt_render_desc* d, int32_t* tri_index_host, float* distance_host);
int rt_primary_hits_device(rt_scene* s, const rt_render_desc* d, int32_t* tri_index_dev, float* distance_dev,
                           void* hip_stream);

/* One process, several GPUs (the C-side form of the spp split of SURVEY 8e; no counterpart in the reference).
 * scenes[k] are replicas of one scene, one per device (rt_init(k); rt_scene_create(...)).  Replica k renders frames
 * [frame_begin + k*frames_each, +frames_each); all devices work at once; sum_host (W*H*3 fp32) += the sum of all
 * n_scenes*frames_each samples per pixel, added in scene order.  Divide by the number of frames accumulated to get
 * lastFrame (equal to the running mean up to fp32 summation order).  Multi-process callers use one process per GPU and
 * one RCCL reduce instead (bench.py). */
int rt_render_multi(rt_scene* const* scenes, int n_scenes, const rt_render_desc* d, int frame_begin, int frames_each,
                    float* sum_host);

/* rt_render_device for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  No counterpart in the reference.  Same arguments, same result (bit-identical to rt_render_device),
 * but it calls no hipMalloc / hipFree / hipEventCreate or other allocating routine: the work-queue counters, the Sobol
 * table, the stack spill of deep BVHs and the sample buffer come out of ONE static 64 MiB device array per GPU (held by
 * every GPU the library runs on until the process ends; its address is looked up per device at rt_init /
 * rt_scene_create).  Limits: 64 KiB header + stack spill (scenes deeper than 18 levels: 2 MiB per extra level) + one
 * frame of samples (width*height*12 B) must fit, i.e. width*height <= 5 586 944 pixels on a shallow scene (1080p,
 * 2560x1440 fit; 3840x2160 does not) -- larger requests return RT_ERR_INVALID_ARG and name the limit.  Frames per
 * launch = what fits (<= rt_tuning.launch_frames), so large images run as more, shorter launches than rt_render_device.
 * All calls on one GPU share that GPU's array (whatever the scene): calls that overlap on one GPU -- two streams, two
 * threads -- race; serialise them.  Kernel timing (rt_kernel_time_ms) does not see these launches. */
int rt_render_device_noalloc(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, float* accum_dev,
                             void* hip_stream);

/* Scheduling knobs of the path loop, per scene (they change how the work is cut, never the image: tested).  Fields <= 0
 * select the default; rt_scene_set_tuning(s, NULL) restores all defaults.  No counterpart in the reference. */
typedef struct rt_tuning {
    int32_t launch_frames;     /* frames per kernel launch (sample buffer = W*H*launch_frames*12 B), max 1024; 0 (default, and what
                                  rt_scene_get_tuning reports for it) = chosen per image size: 2^27 samples per launch, at least 64
                                  frames (1080p: 64, 1280 x 720: 145, 512 x 512: 512) */
    int32_t unit_frames;       /* frames per work unit (one 8x8 tile x unit_frames samples); 0 (default, and what
                                  rt_scene_get_tuning reports for it) = chosen per view: 2; 4 where the scene's root box
                                  covers at most 40 % of the image (rt_scene_screen_coverage), 8 at 15 % or less: mostly sky */
    int32_t leaf_lanes;        /* lanes holding a leaf that trigger a cooperative leaf pass; default 8 (16 on a scene with an entry BVH) */
    int32_t shade_lanes;       /* lanes waiting for the shade pass that trigger it; 0 (default, and what rt_scene_get_tuning
                                  reports for it) = 48; 24 on a scene with an entry BVH; 36 in chapter-3 mode */
    int32_t wait_budget;       /* lane-passes spent waiting that trigger the shade pass; default 4000 */
    int32_t no_cull;           /* > 0: walk every node whose box the ray hits, as the reference does (f5:295-307), instead
                                  of skipping subtrees that provably cannot hold a closer hit; default 0 (skip them) */
    int32_t overlap_frames;    /* launches of fewer frames than this run their path-loop kernel on one of two internal
                                  streams, so that consecutive launches (the reference's one frame per display(), c5:706-760)
                                  overlap: the next one starts while the last paths of this one drain; the fold stays on the
                                  caller's stream.  0 (default, and what rt_scene_get_tuning reports for it) = launches of fewer
                                  than 2^26 samples (1080p: up to 32 frames); 1 = never overlap; 1025 = always.  Overlapping serves ONE
                                  ordinary caller stream per device — the stream of the first rt_render_device /
                                  rt_viewer_display call there; once another stream renders on that device, and for priority
                                  streams, launches run in order on the caller's stream (same images): INTEGRATION.md, "Streams" */
    int32_t no_entry_bvh;      /* > 0: never walk the entry BVH.  Where the caller's BVH is expensive to walk (the reference's
                                  INF = 114514 median splits, c5:500-502, on large scenes: cost_reference_tree > 32) rt_scene_create
                                  builds a private binned-SAH tree over the same triangles; a ray walks it first, with a
                                  conservative box test, to get a distance bound or to prove it hits nothing, and the caller's
                                  BVH — which alone decides the result — only inside that bound.  Same images, fewer nodes. */
    int32_t single_kernel;     /* 1: the whole path loop in one kernel; 2: two kernels per launch — the camera rays of all samples,
                                  coherent work, are traced first by a plain one-lane-per-pixel kernel that also finishes the
                                  samples that escape and leaves 64-byte path states (8.5 GB per 64-frame 1080p launch) from
                                  which the wave-scheduled path loop continues; 0 (default): two kernels for launches of more
                                  than 2^22 samples (width x height x frames), one kernel below — a small launch, the reference's
                                  one frame per display() above all, is all ramp-up and drain.  Same images to the bit. */
} rt_tuning;
int rt_scene_get_tuning(const rt_scene* s, rt_tuning* out);
int rt_scene_set_tuning(rt_scene* s, const rt_tuning* t);

/* What rt_scene_create found.  cull_enabled = 0 when the BVH does not allow closest-hit culling (a box that does not
 * contain its children's boxes or its triangles, a non-finite box) or rt_tuning.no_cull is set: results are the same
 * either way, only the number of nodes visited differs.  No counterpart in the reference. */
typedef struct rt_scene_info {
    int32_t n_triangles, n_nodes;
    int32_t bvh_depth;         /* levels below the root */
    int32_t stack_spill;       /* 1: deeper than the 18 stack entries kept in LDS (global spill in use) */
    int32_t cull_enabled;
    float cull_eps;            /* scene part of a ray's culling margin: 2^-13 * scene extent (each box adds the shape
                                  term of the worst triangle below it: 2^-14 * longest edge / sin(min angle / 2)) */
    int32_t device;
    int32_t entry_bvh;         /* 1: the path loop walks a device-private binned-SAH tree first (see rt_tuning.no_entry_bvh) */
    float cost_reference_tree; /* expected internal nodes a random line through the root box visits: the caller's BVH ... */
    float cost_entry_bvh;      /* ... and the entry BVH (0 when none was built) */
} rt_scene_info;
int rt_scene_get_info(const rt_scene* s, rt_scene_info* out);

/* Share of the image (0..1) that the scene's root box can cover from the descriptor's camera (eye, camera_rotate; the camera
 * model of fshader.fsh:932-937) — an upper bound on the share of camera rays that hit anything; the others take the
 * environment colour at once (fshader.fsh:945).  1 when no estimate is possible (eye inside or beside the box, a
 * camera_rotate that is not a rotation, a BVH whose boxes were not validated).  `unit_frames` (optional) receives the frames
 * per work unit a render call with this descriptor uses (see rt_tuning.unit_frames).  No counterpart in the reference. */
int rt_scene_screen_coverage(const rt_scene* s, const rt_render_desc* d, float* coverage, int* unit_frames);

/* What a render call of `frame_count` frames with this descriptor does on this scene — everything rt_tuning leaves to the
 * library at 0: frames per launch (image size, workspace bound), frames per work unit (view), whether the camera rays get a
 * kernel of their own (launch size), whether consecutive launches overlap (when the caller's stream allows it, see
 * INTEGRATION.md "Streams").  Images never depend on any of it.  No counterpart in the reference. */
typedef struct rt_launch_shape {
    int32_t launch_frames;
    int32_t unit_frames;
    int32_t two_kernels;
    int32_t overlapped;
    float screen_coverage;
} rt_launch_shape;
int rt_scene_launch_shape(const rt_scene* s, const rt_render_desc* d, int frame_count, rt_launch_shape* out);

/* Traversal statistics of a render (same rays as rt_render): totals over all hitBVH invocations. */
typedef struct rt_trace_stats {
    uint64_t paths;            /* pixel samples */
    uint64_t rays;             /* hitBVH invocations (f5:259) */
    uint64_t internal_nodes;   /* internal nodes popped  (I)  */
    uint64_t leaf_nodes;       /* leaf nodes popped      (Lf) */
    uint64_t triangles_tested; /* hitTriangle calls      (T)  */
    uint64_t hit_updates;      /* closest-hit improvements inside hitArray (H), f5:250-253 */
} rt_trace_stats;
int rt_render_stats(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, rt_trace_stats* out);
/* Same counters for the walk the product kernels EXECUTE on these rays: subtrees that provably hold no closer hit are
 * skipped (unless rt_tuning.no_cull) and environment-visibility rays stop at their first hit (only isHit is used,
 * f5:869).  rays and paths equal rt_render_stats'; the other counters are <= it.  This is the work the timed kernel
 * does (bench.py's roofline numerator); rt_render_stats is the reference's algorithmic count. */
int rt_render_stats_executed(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, rt_trace_stats* out);
/* The same counters for the camera rays alone (f5:932-940: one hitBVH per sample; `paths` = the samples that end there, i.e.
 * escape): the share of the walk that the two-kernel form's first kernel does (executed != 0: as executed, else the reference's
 * walk).  RT_ERR_INVALID_ARG where the scene renders with the one-kernel form (see rt_tuning.single_kernel). */
int rt_render_stats_camera(rt_scene* s, const rt_render_desc* d, int frame_begin, int frame_count, int executed, rt_trace_stats* out);

/* Average duration (ms) of the path-loop kernel launches made by the calling thread since the last call with
 * reset != 0, measured with HIP events on the launch stream; *launches receives the count.  Launches of fewer than
 * rt_tuning.overlap_frames frames overlap their neighbours: such a kernel's duration covers a window it shares with the next
 * launch (set overlap_frames = 1 to time a kernel on its own, as bench.py does for its roofline). */
int rt_kernel_time_ms(float* avg_ms, int* launches, int reset);

/* The same with the longest launch and the shader clock the path loop ran at: every wave of the path-loop kernel
 * reports its lifetime in shader-clock ticks (s_memtime); the longest-lived wave spans its launch, so its lifetime /
 * the longest launch's duration = the clock under this load (measured, not the 2.4 GHz of the data sheet).  Call after
 * launches of ONE shape on `s` (the calling thread's timed launches since the last reset are assumed to be that
 * scene's).  Synchronises the device. */
typedef struct rt_kernel_timing_info {
    float avg_ms, max_ms;      /* the path-loop kernel (render_kernel) */
    int32_t launches;
    float shader_clock_mhz;
    float first_avg_ms;        /* two-kernel form: average duration of the camera-ray kernel (first_kernel) that precedes it; else 0 */
} rt_kernel_timing_info;
int rt_kernel_timing(rt_scene* s, rt_kernel_timing_info* out, int reset);

/* ------------------------------------------------------------------------------------------------------------------
 * Viewer semantics: the state and callbacks of the reference's interactive loop, without the window.
 * lastFrame lives on the device (like the reference's texture); nothing is copied to the host until it is read.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct rt_viewer rt_viewer;
/* main() state of c5:147-149, 705: rotatAngle = upAngle = 0, r = 4, frameCounter = 0, a black lastFrame. */
int rt_viewer_create(rt_scene* scene, int width, int height, int mode, rt_viewer** out);
void rt_viewer_destroy(rt_viewer* v);
/* display(), c5:706-760, n_frames times: frames frameCounter .. frameCounter+n-1 into lastFrame, frameCounter += n.
 * Asynchronous on hip_stream (consecutive one-frame calls overlap on the device, rt_tuning.overlap_frames). */
int rt_viewer_display(rt_viewer* v, int n_frames, void* hip_stream);
/* mouseDown / mouse / mouseWheel, c5:766-791: rotatAngle += 150*dx/512, upAngle += 150*dy/512 clamped to +-89,
 * r -= 0.5*direction; every camera change sets frameCounter = 0, so the next frame replaces the picture (weight 1). */
int rt_viewer_mouse_down(rt_viewer* v, double x, double y);
int rt_viewer_mouse_drag(rt_viewer* v, double x, double y);
int rt_viewer_mouse_wheel(rt_viewer* v, int direction);
int rt_viewer_set_orbit(rt_viewer* v, float rotat_angle_deg, float up_angle_deg, float r); /* same, absolute */
int rt_viewer_state(const rt_viewer* v, int* frame_counter, float* rotat_angle_deg, float* up_angle_deg, float* r);
const float* rt_viewer_last_frame_device(const rt_viewer* v);      /* W*H*3 fp32, GL row order, device memory */
int rt_viewer_read(rt_viewer* v, float* rgb_host, void* hip_stream); /* lastFrame -> host; synchronises hip_stream */
int rt_viewer_save_png(rt_viewer* v, const char* path, void* hip_stream); /* pass3 tone-map + gamma, svpng */
/* The sample loop of chapter 1 with its per-N-samples dump (path_tracing_1chap/path_tracing.cpp:407-483, 478-482):
 * renders total_frames more frames and, whenever frameCounter reaches a multiple of dump_every (> 0), writes
 * path_pattern (exactly one %d = the sample count, e.g. "test_path_tracing_spp%d.png").  dump_every = 0: no dumps. */
int rt_viewer_render_progressive(rt_viewer* v, int total_frames, int dump_every, const char* path_pattern, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
