// Host-side scene construction for the MI355X render path.
//
// Mirrors the construction surface of the reference crate -- SceneBuilder (src/scene.rs:8-64),
// MaterialBuilder (src/material.rs:72-131), the Hitable constructors, Camera (src/camera.rs:24-96),
// Config (src/lib.rs:109-147) and TraceSceneBatchJob (src/trace.rs:70-196) -- but instead of a graph of
// Arc<dyn Hitable> it records the graph in a flat arena (the wire format of include/ptgpu.h) that
// pt_scene_upload flattens for the device.  Handles are indices into that arena.
#pragma once

#include "../../include/ptgpu.h"

#include <array>
#include <cstdint>
#include <string>
#include <vector>

namespace ptgpu {

struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    static Vec3 from_float(double f) { return Vec3(f, f, f); }
    static Vec3 new_zero_vector() { return Vec3(0, 0, 0); }
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(Vec3 a, double s) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator*(double s, Vec3 a) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator/(Vec3 a, double s) { return Vec3(a.x / s, a.y / s, a.z / s); }
inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
double dot(Vec3 a, Vec3 b);
Vec3 cross(Vec3 a, Vec3 b);
double length(Vec3 a);
Vec3 unit_vector(Vec3 a);

struct Aabb {
    Vec3 min, max;
};

// Seeded replacement for random::rand (src/math/random.rs:3-6) used while *building* scenes
// (BVH axis choice, random_scene, Perlin tables): a SplitMix64 counter stream.
class SceneRng {
  public:
    explicit SceneRng(uint64_t seed);
    double rand();
    uint64_t draws() const { return n_; }

  private:
    uint64_t key_;
    uint64_t n_ = 0;
};

struct Hitable {
    int id = -1;
};
struct Material {
    int id = -1;
};
struct Texture {
    int id = -1;
};

enum class Axis { X = 0, Y = 1, Z = 2 }; // AxisAlignedRectAxis, src/rect.rs:7-11

// Arena that owns every node, material and texture of one scene.
class Scene {
  public:
    explicit Scene(uint64_t seed);

    // textures (src/texture.rs)
    Texture constant_texture(Vec3 colour);
    Texture checker_texture(Texture even, Texture odd);
    Texture noise_texture(double scale); // builds the Perlin tables on first use (src/math/noise.rs:47-83)
    Texture image_texture(const uint8_t *rgb8, int width, int height);

    // materials (src/material.rs)
    Material lambertian(Texture albedo, double emissive);
    Material diffuse_light(Texture texture);
    Material metal(Vec3 albedo, double fuzz);
    Material dielectric(double ref_idx);
    Material dummy_material();

    // hitables
    Hitable sphere(Vec3 center, double radius, Material m);                                         // sphere.rs:33-41
    Hitable moving_sphere(Vec3 c0, Vec3 c1, double t0, double t1, double radius, Material m);       // sphere.rs:115-128
    Hitable rect(double amin, double amax, double bmin, double bmax, double c, Axis axis, Material m); // rect.rs:28-61
    Hitable axis_aligned_box(Vec3 pmin, Vec3 pmax, Material m);                                     // axis_aligned_box.rs:14-29
    Hitable flip_normals(Hitable child);                                                            // hitable.rs:84-90
    Hitable translate(Hitable child, Vec3 translation);                                             // hitable.rs:112-119
    Hitable rotate_y(Hitable child, double angle_degrees);                                          // hitable.rs:147-180
    Hitable constant_medium(Hitable boundary, double density, Texture texture);                     // volume.rs:13-22
    Hitable hitable_list(const std::vector<Hitable> &list);                                         // hitable.rs:44-48
    Hitable bvh_from_list(std::vector<Hitable> list, double time0, double time1);                   // bvh.rs:13-69

    // bounding_box(t0, t1) of a node; false for a list (unreachable!(), hitable.rs:63-65)
    bool bounding_box(Hitable h, double t0, double t1, Aabb &out) const;

    void set_root(Hitable h) { root_ = h.id; }
    void set_lights(const std::vector<Hitable> &shapes);
    // The hard-coded light-shape list of trace.rs:123-128: Cornell light rect + glass sphere, dummy material.
    void set_default_lights();

    SceneRng &rng() { return rng_; }
    pt_scene_desc desc() const;
    int n_hitables() const { return (int)hitables_.size(); }
    const pt_hitable &node(int i) const { return hitables_[i]; }
    const std::string &error() const { return error_; }

  private:
    int push(const pt_hitable &h);
    void fail(const std::string &msg);

    SceneRng rng_;
    std::vector<pt_hitable> hitables_;
    std::vector<int32_t> child_index_;
    std::vector<pt_material> materials_;
    std::vector<pt_texture> textures_;
    std::vector<uint8_t> texels_;
    std::vector<int32_t> lights_;
    pt_perlin perlin_;
    bool have_perlin_ = false;
    int root_ = -1;
    std::string error_;
};

// MaterialBuilder, src/material.rs:72-131
class MaterialBuilder {
  public:
    explicit MaterialBuilder(Scene &scene);
    MaterialBuilder &with_texture(Texture t);
    MaterialBuilder &set_emissive(double e);
    MaterialBuilder &set_albedo(Vec3 a);
    MaterialBuilder &set_fuzz(double f);
    MaterialBuilder &set_refraction_index(double ri);
    Material lambertian();
    Material diffuse_light();
    Material metal();
    Material dielectric();

  private:
    Scene &scene_;
    Texture texture_;
    Vec3 albedo_;
    double emissive_ = 0.0, fuzz_ = 0.0, refraction_index_ = 1.0;
};

// SceneBuilder, src/scene.rs:8-64: a stack of hitables whose wrappers re-wrap the last pushed one.
class SceneBuilder {
  public:
    explicit SceneBuilder(Scene &scene) : scene_(scene) {}
    SceneBuilder &add_hitable(Hitable h);
    SceneBuilder &flip_normals();
    SceneBuilder &translate(Vec3 translation);
    SceneBuilder &rotate_y(double angle_degrees);
    Hitable as_bvh();          // BvhNode::from_list(scene, 0.0, 1.0)
    Hitable as_bvh_node();     // same node, usable as a child
    Hitable as_hitable_list();
    Hitable as_hitable();      // pops the last pushed hitable

  private:
    Scene &scene_;
    std::vector<Hitable> list_;
};

// Camera, src/camera.rs:6-107 (the interactive update_from_input is UI and not mirrored)
class Camera {
  public:
    Camera(Vec3 origin, Vec3 look_at, Vec3 vup, double vfov, double aspect, double aperture, double focus_dist,
           double time0, double time1);
    Vec3 get_forward() const { return -w_; }
    Vec3 get_up() const { return v_; }
    Vec3 get_right() const { return u_; }
    Vec3 get_origin() const { return origin_; }
    Vec3 get_look_at() const { return look_at_; }
    void set_origin(Vec3 origin, bool update_look_at);
    void set_look_at(Vec3 look_at, bool maintain_distance);
    void update();
    // camera.rs:109-215: WASDQE move the origin (and the look-at with it), the arrow keys and the mouse move the
    // look-at at constant distance. Returns camera_moved. `input` is a pt_user_input (include/ptgpu.h).
    bool update_from_input(const pt_user_input &input, double frame_time);
    Vec3 lower_left_corner() const { return lower_left_corner_; }
    Vec3 horizontal() const { return horizontal_; }
    Vec3 vertical() const { return vertical_; }
    // The Camera::new parameter set that reproduces the current state on the device.
    pt_camera params() const;

  private:
    Vec3 origin_, look_at_, lower_left_corner_, horizontal_, vertical_, u_, v_, w_, world_up_;
    double time0_, time1_, lens_radius_, half_width_, half_height_, focus_dist_, vfov_, aspect_;
};

// Config, src/lib.rs:109-147
struct Config {
    bool realtime = true;
    int max_depth = 10;
    uint32_t spp = 1;
    static Config from_cmdline(const std::vector<std::string> &args);
};

// Scene recipes (src/lib.rs:674-1077) and the synthetic sphere cloud. Each sets root and lights.
struct BuiltScene {
    Scene *scene = nullptr;
    pt_camera camera{};
};
// Decoded RGB8 texels of an image file a scene recipe refers to (rows top-down as decoded), kept for the process.
bool register_image_asset(const std::string &name, const uint8_t *rgb8, int width, int height);
bool build_named_scene(const std::string &name, double aspect, uint64_t seed, int64_t param, Scene &scene,
                       pt_camera &camera, std::string &error);

// SceneState / SceneOutput / TraceSceneBatchJob, src/trace.rs:22-196: one tile job bound to a device context.
struct SceneOutput {
    pt_ctx *ctx = nullptr;   // the image lives in the context (SceneOutput.buffer)
    uint32_t remaining_tasks = 0;
    void notify_task_completion() { if (remaining_tasks) --remaining_tasks; }
};
struct SceneState {
    pt_ctx *ctx = nullptr;
    Config config;
    uint32_t frame = 0;
};
class TraceSceneBatchJob {
  public:
    TraceSceneBatchJob(uint32_t num_samples, std::array<uint32_t, 2> start_xy, std::array<uint32_t, 2> end_xy,
                       std::array<uint32_t, 2> image_size, SceneState *state, SceneOutput *output, bool realtime);
    int run(); // JobTask::run -> trace(); returns a pt_status
    void clear_buffer();
    pt_render_args render_args() const; // this job as the C ABI takes it
    pt_ctx *context() const { return state_->ctx; }
    SceneOutput *output() const { return output_; }

  private:
    uint32_t num_samples_;
    std::array<uint32_t, 2> start_xy_, end_xy_, image_size_;
    SceneState *state_;
    SceneOutput *output_;
    bool realtime_;
};

// Jobs::dispatch_jobs + wait_for_counter (src/lib.rs:396-399,476-478) for a batch of tile jobs bound to one context:
// the batch goes to the library in one call, which traces a batch that tiles a rectangle as one pass (pt_render_jobs).
int dispatch_jobs(std::vector<TraceSceneBatchJob> &jobs);

// The body of the realtime main loop (src/lib.rs:406-478) without the window: per frame it advances the scene time,
// applies the exposure / sky / emissive keys and the camera input, clears the accumulation when something moved,
// and traces one progressive frame (spp 1, depth 10) into the context's image.
class RealtimeSession {
  public:
    RealtimeSession(pt_ctx *ctx, const pt_camera &camera, uint32_t width, uint32_t height, uint32_t max_depth);
    int frame(const pt_user_input &input, double frame_time, pt_frame_info *info); // returns a pt_status
    const Camera &camera() const { return cam_; }

  private:
    pt_ctx *ctx_;
    Camera cam_;
    uint32_t width_, height_, max_depth_;
    double time0_ = 0.0, time1_ = 1.0; // SceneState.time0/1 (advanced every frame, read by nothing: lib.rs:415-416)
    double exposure_ = 1.3, numframes_ = 1.0; // TonemapperArgs.exposure_numframes_xx, lib.rs:284-286
    double sky_brightness_ = 0.0;             // stored and toggled, never read by color (trace.rs:49-50,227)
    bool disable_emissive_ = false;
    uint32_t frame_counter_ = 0;
};

// Tile grid of lib.rs:345-351, extended so edge tiles cover the whole image.
std::vector<std::array<uint32_t, 4>> make_tile_grid(uint32_t nx, uint32_t ny);

} // namespace ptgpu

// Owner of a Scene across the C ABI.
struct pt_scene {
    ptgpu::Scene scene;
    explicit pt_scene(uint64_t seed) : scene(seed) {}
};
