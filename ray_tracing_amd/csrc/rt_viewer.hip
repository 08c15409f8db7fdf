// rt_viewer: the state and the callbacks of the reference's interactive loop (path_tracing_5chap.cpp, abbrev. c5) without the
// window — SURVEY 8(f)3.  display() (c5:706-760) = one frame per call into a device-resident lastFrame with frameCounter++;
// mouse() (c5:766-776) and mouseWheel() (c5:786-791) change the orbit camera and reset frameCounter to 0, which makes the next
// frame replace the picture (mix weight 1/(0+1), f5:954-955); the per-N-samples PNG dump is chapter 1's commented-out loop
// (path_tracing_1chap/path_tracing.cpp:478-482).  Host code only: it drives the C ABI of rt_device.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <string>
#include <vector>

#include "rt_host.h"

using namespace rta;

struct rt_viewer {
    rt_scene* scene = nullptr;
    rt_render_desc desc{};
    float rotat_angle = 0.0f, up_angle = 0.0f, r = 4.0f;  // c5:147-149
    double last_x = 0.0, last_y = 0.0;                    // c5:765
    int frame_counter = 0;                                // c5:705
    int device = 0;
    float* d_last_frame = nullptr;                        // lastFrame, W*H*3 fp32 on the device
    std::vector<float> host;                              // read-back staging for PNG dumps
};

namespace {

int update_camera(rt_viewer* v) { return rt_camera(v->rotat_angle, v->up_angle, v->r, v->desc.eye, v->desc.camera_rotate); }

struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev); else prev = -1;
    }
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

extern "C" {

int rt_viewer_create(rt_scene* scene, int width, int height, int mode, rt_viewer** out) {
    if (!out) return set_error(RT_ERR_INVALID_ARG, "rt_viewer_create: out is NULL");
    *out = nullptr;
    if (!scene) return set_error(RT_ERR_INVALID_ARG, "rt_viewer_create: NULL scene");
    rt_scene_info info;
    int rc = rt_scene_get_info(scene, &info);
    if (rc != RT_OK) return rc;
    rt_viewer* v = new rt_viewer();
    v->scene = scene;
    v->device = info.device;
    v->desc.width = width, v->desc.height = height, v->desc.mode = mode;
    v->desc.max_bounce = 0;                         // the shader's literal (f5:949, f4:564)
    v->desc.accumulate = RT_ACCUM_RUNNING_MEAN;     // f5:954-955
    rc = update_camera(v);
    if (rc == RT_OK && (width <= 0 || height <= 0 || width > 65535 || height > 32767))
        rc = set_error(RT_ERR_INVALID_ARG, "rt_viewer_create: bad image size %dx%d", width, height);
    if (rc != RT_OK) {
        delete v;
        return rc;
    }
    DeviceScope scope(v->device);
    const size_t bytes = (size_t)width * height * 3 * sizeof(float);
    if (hipMalloc((void**)&v->d_last_frame, bytes) != hipSuccess || hipMemset(v->d_last_frame, 0, bytes) != hipSuccess) {
        (void)hipFree(v->d_last_frame);
        delete v;
        return set_error(RT_ERR_OOM, "rt_viewer_create: cannot allocate the %zu-byte lastFrame", bytes);
    }
    *out = v;
    return RT_OK;
}

void rt_viewer_destroy(rt_viewer* v) {
    if (!v) return;
    DeviceScope scope(v->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(v->d_last_frame);
    delete v;
}

int rt_viewer_set_orbit(rt_viewer* v, float rotat_angle_deg, float up_angle_deg, float r) {
    if (!v) return set_error(RT_ERR_INVALID_ARG, "rt_viewer_set_orbit: NULL viewer");
    v->rotat_angle = rotat_angle_deg, v->up_angle = up_angle_deg, v->r = r;
    v->frame_counter = 0;
    return update_camera(v);
}

int rt_viewer_mouse_down(rt_viewer* v, double x, double y) {  // c5:778-784
    if (!v) return set_error(RT_ERR_INVALID_ARG, "rt_viewer_mouse_down: NULL viewer");
    v->last_x = x, v->last_y = y;
    return RT_OK;
}

int rt_viewer_mouse_drag(rt_viewer* v, double x, double y) {  // mouse(), c5:766-776
    if (!v) return set_error(RT_ERR_INVALID_ARG, "rt_viewer_mouse_drag: NULL viewer");
    v->frame_counter = 0;
    v->rotat_angle += (float)(150 * (x - v->last_x) / 512);
    v->up_angle += (float)(150 * (y - v->last_y) / 512);
    v->up_angle = fminf(v->up_angle, 89.0f);
    v->up_angle = fmaxf(v->up_angle, -89.0f);
    v->last_x = x, v->last_y = y;
    return update_camera(v);
}

int rt_viewer_mouse_wheel(rt_viewer* v, int direction) {  // mouseWheel(), c5:786-791
    if (!v) return set_error(RT_ERR_INVALID_ARG, "rt_viewer_mouse_wheel: NULL viewer");
    v->frame_counter = 0;
    v->r += (float)(-direction * 0.5);
    return update_camera(v);
}

int rt_viewer_display(rt_viewer* v, int n_frames, void* hip_stream) {  // display() x n_frames, c5:706-760
    if (!v) return set_error(RT_ERR_INVALID_ARG, "rt_viewer_display: NULL viewer");
    if (n_frames < 0) return set_error(RT_ERR_INVALID_ARG, "rt_viewer_display: negative frame count");
    const int rc = rt_render_device(v->scene, &v->desc, v->frame_counter, n_frames, v->d_last_frame, hip_stream);
    if (rc == RT_OK) v->frame_counter += n_frames;  // frameCounter++, c5:729
    return rc;
}

int rt_viewer_state(const rt_viewer* v, int* frame_counter, float* rotat_angle_deg, float* up_angle_deg, float* r) {
    if (!v) return set_error(RT_ERR_INVALID_ARG, "rt_viewer_state: NULL viewer");
    if (frame_counter) *frame_counter = v->frame_counter;
    if (rotat_angle_deg) *rotat_angle_deg = v->rotat_angle;
    if (up_angle_deg) *up_angle_deg = v->up_angle;
    if (r) *r = v->r;
    return RT_OK;
}

const float* rt_viewer_last_frame_device(const rt_viewer* v) { return v ? v->d_last_frame : nullptr; }

int rt_viewer_read(rt_viewer* v, float* rgb_host, void* hip_stream) {
    if (!v || !rgb_host) return set_error(RT_ERR_INVALID_ARG, "rt_viewer_read: NULL argument");
    DeviceScope scope(v->device);
    const size_t bytes = (size_t)v->desc.width * v->desc.height * 3 * sizeof(float);
    hipError_t e = hipMemcpyAsync(rgb_host, v->d_last_frame, bytes, hipMemcpyDeviceToHost, (hipStream_t)hip_stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)hip_stream);
    if (e != hipSuccess) return set_error(RT_ERR_HIP, "rt_viewer_read: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_viewer_save_png(rt_viewer* v, const char* path, void* hip_stream) {  // pass3 (pass3.fsh:14-25) + svpng
    if (!v || !path) return set_error(RT_ERR_INVALID_ARG, "rt_viewer_save_png: NULL argument");
    const size_t n = (size_t)v->desc.width * v->desc.height * 3;
    v->host.resize(n);
    int rc = rt_viewer_read(v, v->host.data(), hip_stream);
    if (rc != RT_OK) return rc;
    std::vector<unsigned char> rgb8(n);
    rc = rt_tonemap_rgb8(v->host.data(), v->desc.width, v->desc.height, rgb8.data(), 1);
    if (rc != RT_OK) return rc;
    return rt_write_png(path, v->desc.width, v->desc.height, rgb8.data(), 0);
}

int rt_viewer_render_progressive(rt_viewer* v, int total_frames, int dump_every, const char* path_pattern, void* hip_stream) {
    // chapter 1's loop, c1:407-483 with the dump of c1:478-482: "every `dump_every` samples write test_path_tracing_spp<k>.png"
    if (!v) return set_error(RT_ERR_INVALID_ARG, "rt_viewer_render_progressive: NULL viewer");
    if (total_frames < 0 || dump_every < 0 || (dump_every > 0 && !path_pattern))
        return set_error(RT_ERR_INVALID_ARG, "rt_viewer_render_progressive: bad arguments");
    if (dump_every > 0) {  // exactly one %d, nothing else: the pattern goes through snprintf
        int conv = 0;
        for (const char* p = path_pattern; *p; ++p)
            if (*p == '%') {
                if (p[1] == 'd' && conv == 0) ++conv, ++p;
                else return set_error(RT_ERR_INVALID_ARG, "rt_viewer_render_progressive: the path pattern takes exactly one %%d (the sample count)");
            }
        if (conv != 1) return set_error(RT_ERR_INVALID_ARG, "rt_viewer_render_progressive: the path pattern takes exactly one %%d (the sample count)");
    }
    int done = 0;
    while (done < total_frames) {
        int n = total_frames - done;
        if (dump_every > 0) {
            const int to_dump = dump_every - (v->frame_counter % dump_every);
            if (to_dump < n) n = to_dump;
        }
        int rc = rt_viewer_display(v, n, hip_stream);
        if (rc != RT_OK) return rc;
        done += n;
        if (dump_every > 0 && v->frame_counter % dump_every == 0) {
            char path[4096];
            if (snprintf(path, sizeof(path), path_pattern, v->frame_counter) >= (int)sizeof(path))
                return set_error(RT_ERR_INVALID_ARG, "rt_viewer_render_progressive: path too long");
            rc = rt_viewer_save_png(v, path, hip_stream);
            if (rc != RT_OK) return rc;
        }
    }
    return RT_OK;
}

}  // extern "C"
