// Host layer under AddressSanitizer + UBSan, no GPU: builds every scene recipe, flattens it and builds its SAH tree
// (tools/host_sanitize.sh compiles this with the host sources of the library).
#include "ptgpu.h"
#include <cstdio>
#include <vector>
int main() {
    const char *names[] = {"cornell_box", "cornell_smoke", "random_scene", "final_book_two", "two_spheres", "four_spheres",
                           "two_perlin_spheres", "simple_light", "sphere_cloud", "textured_sphere"};
    { std::vector<uint8_t> img(1024 * 512 * 3, 128); pt_asset_register_image("earthmap.jpg", img.data(), 1024, 512); }
    for (const char *n : names) {
        pt_scene *s = nullptr; pt_camera cam;
        int rc = pt_scene_build_named(n, 1.0, 0x5EED0001ull, 20000, &s, &cam);
        if (rc != PT_OK) { std::printf("%s: build rc %d\n", n, rc); continue; }
        pt_scene_desc d; pt_scene_get_desc(s, &d);
        int32_t a, b, c, e;
        rc = pt_scene_flatten_info(&d, &a, &b, &c, &e);
        int32_t nn = 0, np = 0, dep = 0;
        int rc2 = pt_scene_accel_info(&d, &nn, &np, &dep);
        std::printf("%s: flatten rc %d ops %d leaves %d; accel rc %d nodes %d prims %d depth %d\n", n, rc, a, b, rc2, nn, np, dep);
        pt_scene_free(s);
    }
    return 0;
}
// device entry points the host layer refers to: never reached by this driver
extern "C" int pt_accum_reset(pt_ctx *) { return PT_ERR_NO_DEVICE; }
extern "C" int pt_render(pt_ctx *, const pt_render_args *) { return PT_ERR_NO_DEVICE; }
extern "C" int pt_render_jobs(pt_ctx *, const pt_render_args *, size_t) { return PT_ERR_NO_DEVICE; }
extern "C" int pt_camera_set(pt_ctx *, const pt_camera *) { return PT_ERR_NO_DEVICE; }
