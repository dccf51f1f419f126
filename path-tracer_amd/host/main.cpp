// Command-line front end mirroring src/main.rs:7-24 and the offline branch of path_tracer::run
// (src/lib.rs:163-575) without the window: `-offline` and `-spp=N` are the reference's flags
// (src/lib.rs:125-146); the scene is the Cornell box at 500x500 (src/lib.rs:173-174,328) unless overridden.
// Writes output0.ppm with the Reinhard + gamma conversion of src/lib.rs:330-340,613-640.
#include "ptscene.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace ptgpu;

int main(int argc, char **argv) {
    std::vector<std::string> args(argv + 1, argv + argc);
    Config config = Config::from_cmdline(args);
    uint32_t nx = 500, ny = 500;
    std::string scene_name = "cornell_box", out = "output0.ppm";
    for (auto &a : args) {
        if (a.rfind("-width=", 0) == 0) nx = (uint32_t)std::atoi(a.c_str() + 7);
        if (a.rfind("-height=", 0) == 0) ny = (uint32_t)std::atoi(a.c_str() + 8);
        if (a.rfind("-scene=", 0) == 0) scene_name = a.substr(7);
        if (a.rfind("-out=", 0) == 0) out = a.substr(5);
        // decoded texels of assets/textures/earthmap.jpg: 1024 x 512 RGB8, rows top-down (src/lib.rs:767)
        if (a.rfind("-earthmap=", 0) == 0) {
            std::vector<uint8_t> texels((size_t)1024 * 512 * 3);
            FILE *fp = std::fopen(a.c_str() + 10, "rb");
            const size_t got = fp ? std::fread(texels.data(), 1, texels.size(), fp) : 0;
            if (fp) std::fclose(fp);
            if (got != texels.size() || pt_asset_register_image("earthmap.jpg", texels.data(), 1024, 512) != PT_OK) {
                std::fprintf(stderr, "%s: expected %zu bytes of RGB8 texels\n", a.c_str() + 10, texels.size());
                return 1;
            }
        }
    }
    std::printf("realtime: %d max_depth: %d spp: %u\n", config.realtime ? 1 : 0, config.max_depth, config.spp);

    pt_scene *scene = nullptr;
    pt_camera cam;
    const int built = pt_scene_build_named(scene_name.c_str(), (double)nx / (double)ny, 0x5EED0001ull, 0, &scene, &cam);
    if (built != PT_OK) {
        std::fprintf(stderr, built == PT_ERR_MISSING_ASSET ? "scene %s needs -earthmap=FILE (RGB8 texels of earthmap.jpg)\n"
                                                           : "unknown scene %s\n", scene_name.c_str());
        return 1;
    }
    pt_config cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.seed = 0x5EED0001ull;
    pt_ctx *ctx = nullptr;
    int rc = pt_create(&cfg, &ctx);
    if (rc != PT_OK) {
        std::fprintf(stderr, "pt_create failed: %d (no MI355X visible?)\n", rc);
        return 1;
    }
    pt_scene_desc desc;
    pt_scene_get_desc(scene, &desc);
    if ((rc = pt_scene_upload(ctx, &desc)) != PT_OK || (rc = pt_camera_set(ctx, &cam)) != PT_OK) {
        std::fprintf(stderr, "scene upload failed: %d %s\n", rc, pt_last_error(ctx));
        return 1;
    }
    SceneState state;
    state.ctx = ctx;
    state.config = config;
    SceneOutput output;
    output.ctx = ctx;
    // tile jobs as in src/lib.rs:345-394, dispatched as one batch per frame (src/lib.rs:396-399)
    auto tiles = make_tile_grid(nx, ny);
    output.remaining_tasks = (uint32_t)tiles.size();
    std::printf("Tracing... %zu tasks\n", tiles.size());
    const int frames = config.realtime ? 16 : 1;
    auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; ++f) {
        state.frame = (uint32_t)f;
        std::vector<TraceSceneBatchJob> jobs;
        for (auto &t : tiles)
            jobs.emplace_back(config.spp, std::array<uint32_t, 2>{t[0], t[1]}, std::array<uint32_t, 2>{t[2], t[3]},
                              std::array<uint32_t, 2>{nx, ny}, &state, &output, config.realtime);
        if ((rc = dispatch_jobs(jobs)) != PT_OK) {
            std::fprintf(stderr, "render failed: %d %s\n", rc, pt_last_error(ctx));
            return 1;
        }
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("Done.. in %fs.\n", secs); // src/lib.rs:485-487
    std::vector<float> rgba((size_t)nx * ny * 4);
    pt_readback(ctx, rgba.data(), rgba.size());
    std::vector<uint8_t> rgb((size_t)nx * ny * 3);
    pt_tonemap_rgb8(rgba.data(), nx, ny, rgb.data());
    if (FILE *fp = std::fopen(out.c_str(), "wb")) {
        std::fprintf(fp, "P6 %u %u 255\n", nx, ny);
        std::fwrite(rgb.data(), 1, rgb.size(), fp);
        std::fclose(fp);
        std::printf("%s saved\n", out.c_str());
    }
    pt_destroy(ctx);
    pt_scene_free(scene);
    return 0;
}
