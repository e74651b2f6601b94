// The reference's C++ class API (s2899221/Code/{utils,persp_camera,mesh,raytracer}.h) on top of librt_hip.so.
//
// Same class names, public signatures, ownership and error behaviour as the reference, so its callers
// (s2899221/Code/render_driver.cpp:31-40, tests/raytracer/time_driver.cpp:33-38) compile unchanged; every
// intersection, shading and render call runs on the GPU through the C ABI of include/rt_hip.h.  There is no
// CPU tracing path: without a usable GPU these calls throw std::runtime_error.
//
// Additions over the reference are marked "extension": batch overloads (one GPU launch for many rays), the
// Triangle primitive, and explicit seeds for multi-sample renders.
#ifndef RT_HOST_HPP
#define RT_HOST_HPP

#include <iosfwd>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "rt_hip.h"

// ---- utils.h ---------------------------------------------------------------------------------------------

namespace util_json {
// utils.h:14-23.  Text scanning helpers of the reference's loader, restated without <regex> (linear time).
bool extract_object_block(const std::string& src, const std::string& key, std::string& out_block);
bool parse_number(const std::string& src, const std::string& key, double& out);
bool parse_vec3(const std::string& src, const std::string& key, double& x, double& y, double& z);
bool parse_vec2i(const std::string& src, const std::string& key, int& outx, int& outy);
}

class Mesh;
class BoundingBox;
class Image;

struct RayVec3 { double x, y, z; };     // utils.h:31
struct HitVec3 { double x, y, z; };     // utils.h:61
struct Pixel { unsigned char r, g, b; };   // utils.h:122

// utils.h:38-56
class Ray {
public:
    Ray() : m_o{ 0, 0, 0 }, m_d{ 0, 0, 1 } {}
    Ray(const RayVec3& position, const RayVec3& direction) : m_o(position), m_d(direction) {}
    const RayVec3& getPosition() const { return m_o; }
    const RayVec3& getDirection() const { return m_d; }
    void setPosition(const RayVec3& p) { m_o = p; }
    void setDirection(const RayVec3& d) { m_d = d; }
private:
    RayVec3 m_o, m_d;
};

// utils.h:68-117.  One flat record plus a presence mask instead of seven std::optional members; getters throw
// std::runtime_error("Hit: <field> is unset") exactly when the reference's do.
class Hit {
public:
    Hit(std::optional<HitVec3> intersectionPoint = std::nullopt, std::optional<double> distanceAlongRay = std::nullopt,
        std::optional<HitVec3> surfaceNormal = std::nullopt, std::optional<HitVec3> reflectedDirection = std::nullopt,
        std::optional<const Mesh*> mesh = std::nullopt);

    const HitVec3& getIntersectionPoint() const { need(kPoint, "intersectionPoint"); return m_p; }
    double getDistanceAlongRay() const { need(kDist, "distanceAlongRay"); return m_t; }
    const HitVec3& getSurfaceNormal() const { need(kNormal, "surfaceNormal"); return m_n; }
    const HitVec3& getReflectedDirection() const { need(kReflect, "reflectedDirection"); return m_r; }
    const Mesh* getMesh() const { need(kMesh, "mesh"); return m_mesh; }

    void setIntersectionPoint(const HitVec3& p) { m_p = p; m_has |= kPoint; }
    void setDistanceAlongRay(double t) { m_t = t; m_has |= kDist; }
    void setSurfaceNormal(const HitVec3& n) { m_n = n; m_has |= kNormal; }
    void setReflectedDirection(const HitVec3& r) { m_r = r; m_has |= kReflect; }
    void setMesh(const Mesh* m) { m_mesh = m; m_has |= kMesh; }

    bool hasIntersectionPoint() const { return m_has & kPoint; }
    bool hasDistanceAlongRay() const { return m_has & kDist; }
    bool hasSurfaceNormal() const { return m_has & kNormal; }
    bool hasReflectedDirection() const { return m_has & kReflect; }
    bool hasMesh() const { return m_has & kMesh; }

    void setUV(double u, double v) { m_u = u; m_v = v; m_has |= kUV; }
    bool hasUV() const { return m_has & kUV; }
    double getU() const { need(kUV, "u"); return m_u; }
    double getV() const { need(kUV, "v"); return m_v; }

    // extension: fill from / test against one 16-double record of rt_trace_rays_full
    static Hit from_record(const double* rec16, const Mesh* mesh);
private:
    enum : unsigned { kPoint = 1, kDist = 2, kNormal = 4, kReflect = 8, kMesh = 16, kUV = 32 };
    void need(unsigned bit, const char* name) const {
        if (!(m_has & bit)) throw std::runtime_error(std::string("Hit: ") + name + " is unset");
    }
    HitVec3 m_p{ 0, 0, 0 }, m_n{ 0, 0, 0 }, m_r{ 0, 0, 0 };
    double m_t = 0.0, m_u = 0.0, m_v = 0.0;
    const Mesh* m_mesh = nullptr;
    unsigned m_has = 0;
};

// utils.h:129-151: 8-bit PPM image (P3 / P6 in, P6 out).
class Image {
public:
    explicit Image(const std::string& filename);
    Image(int width, int height);
    int width() const { return m_w; }
    int height() const { return m_h; }
    int maxValue() const { return m_max; }
    Pixel getPixel(int x, int y) const;
    void setPixel(int x, int y, int r, int g, int b);
    void addToPixel(int x, int y, int dr, int dg, int db);
    void setMaxValue(int maxVal);
    void write(const std::string& filename) const;
    // extension: bulk access for frames that arrive from the GPU as packed RGB
    void assign_rgb(const unsigned char* rgb);
    const std::vector<Pixel>& pixels() const { return m_px; }
private:
    int m_w, m_h, m_max;
    std::vector<Pixel> m_px;
};

// ---- persp_camera.h ---------------------------------------------------------------------------------------

struct Vec3 {                               // persp_camera.h:7-18
    double x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(double _x, double _y, double _z) : x(_x), y(_y), z(_z) {}
    Vec3 operator-(const Vec3& v) const { return Vec3(x - v.x, y - v.y, z - v.z); }
    Vec3 operator+(const Vec3& v) const { return Vec3(x + v.x, y + v.y, z + v.z); }
    Vec3 operator*(double s) const { return Vec3(x * s, y * s, z * s); }
    double length() const;
    Vec3 normalized() const;
};

// persp_camera.h:28-61.  State is the rt_camera the GPU entry points take.
class Camera {
public:
    Camera();
    void setPosition(const Vec3& pos);
    void setFocalLength(double focal_mm) { m_cam.focal = focal_mm; }
    void setSensorSize(double width_mm, double height_mm) { m_cam.sensor_w = width_mm; m_cam.sensor_h = height_mm; }
    void setImageResolution(int width_px, int height_px) { m_cam.width = width_px; m_cam.height = height_px; }
    void getImageResolution(int& width_px, int& height_px) const { width_px = m_cam.width; height_px = m_cam.height; }
    void updateRotationMatrix(const Vec3& cam_pos, const Vec3& cam_dir);
    void updateRotationMatrix(const Vec3& cam_pos, const Vec3& cam_dir, const Vec3& cam_up);
    bool worldToPixel(const Vec3& point_world, int& pixel_x, int& pixel_y);
    void pixelToRay(double pixel_x, double pixel_y, Vec3& ray_origin, Vec3& ray_direction) const;
    // extension
    const rt_camera& raw() const { return m_cam; }
    explicit Camera(const rt_camera& c) : m_cam(c) {}
private:
    rt_camera m_cam;
};

bool read_camera_from_json(const std::string& json_filepath, const std::string& camera_id, Camera& camera);   // persp_camera.h:66

// ---- mesh.h -----------------------------------------------------------------------------------------------

struct Float3 { double x, y, z; };                        // mesh.h:64
struct EulerAngles { double roll, pitch, yaw; };          // mesh.h:71

// mesh.h:15-61.  Move-only (owns its texture); the geometry and material live in one rt_prim_desc.
class Mesh {
public:
    Mesh();
    virtual ~Mesh();
    Mesh(const Mesh&) = delete;
    Mesh& operator=(const Mesh&) = delete;
    Mesh(Mesh&&) noexcept;
    Mesh& operator=(Mesh&&) noexcept;

    virtual void write_to_console(std::ostream& out) const = 0;
    virtual bool intersect(const Ray& ray, Hit& hit) const;          // base: no geometry, false (mesh.h:28)
    virtual void compute_aabb(Float3& outMin, Float3& outMax) const = 0;
    virtual const BoundingBox* asBoundingBox() const { return nullptr; }

    void setMaterial(double alpha, double metallic, double roughness) { m_desc.alpha = alpha; m_desc.metallic = metallic; m_desc.roughness = roughness; }
    double getAlpha() const { return m_desc.alpha; }
    double getMetallic() const { return m_desc.metallic; }
    double getRoughness() const { return m_desc.roughness; }
    void setIndexOfRefraction(double ior) { m_desc.ior = ior; }
    double getIndexOfRefraction() const { return m_desc.ior; }
    void setAlbedo(double r, double g, double b) { m_desc.albedo[0] = r; m_desc.albedo[1] = g; m_desc.albedo[2] = b; }
    void setAlbedoTexture(std::unique_ptr<Image> texture);
    virtual Pixel evaluate_albedo(const Hit& hit) const;

    // extension: the description the C ABI takes (kind < 0 for a BoundingBox), the texture if any, and the
    // primitive's intersect for many rays in one launch
    const rt_prim_desc& describe() const { return m_desc; }
    const Image* albedoTexture() const { return m_albedoTexture.get(); }
    std::vector<Hit> intersect(const std::vector<Ray>& rays) const;
protected:
    rt_prim_desc m_desc;
    std::unique_ptr<Image> m_albedoTexture;
    bool gpu_intersect(const Ray& ray, Hit& hit) const;
    void gpu_aabb(Float3& outMin, Float3& outMax) const;
};

#define RT_HOST_MOVE_ONLY(T) \
    T(const T&) = delete; T& operator=(const T&) = delete; T(T&&) noexcept = default; T& operator=(T&&) noexcept = default;

class Cube : public Mesh {                                 // mesh.h:78-98
public:
    Cube();
    Cube(const Float3& translation, const EulerAngles& rotation, const Float3& scale);
    RT_HOST_MOVE_ONLY(Cube)
    static std::vector<Cube> read_from_json(const std::string& class_block);
    void write_to_console(std::ostream& out) const override;
    bool intersect(const Ray& ray, Hit& hit) const override { return gpu_intersect(ray, hit); }
    using Mesh::intersect;
    void compute_aabb(Float3& outMin, Float3& outMax) const override { gpu_aabb(outMin, outMax); }
};

class Cylinder : public Mesh {                             // mesh.h:104-123
public:
    Cylinder();
    Cylinder(const Float3& translation, const EulerAngles& rotation, const Float3& scale);
    RT_HOST_MOVE_ONLY(Cylinder)
    static std::vector<Cylinder> read_from_json(const std::string& class_block);
    void write_to_console(std::ostream& out) const override;
    bool intersect(const Ray& ray, Hit& hit) const override { return gpu_intersect(ray, hit); }
    using Mesh::intersect;
    void compute_aabb(Float3& outMin, Float3& outMax) const override { gpu_aabb(outMin, outMax); }
};

class Sphere : public Mesh {                               // mesh.h:129-147
public:
    Sphere();
    Sphere(const Float3& location, const Float3& scale);
    RT_HOST_MOVE_ONLY(Sphere)
    static std::vector<Sphere> read_from_json(const std::string& class_block);
    void write_to_console(std::ostream& out) const override;
    bool intersect(const Ray& ray, Hit& hit) const override { return gpu_intersect(ray, hit); }
    using Mesh::intersect;
    void compute_aabb(Float3& outMin, Float3& outMax) const override { gpu_aabb(outMin, outMax); }
};

class Plane : public Mesh {                                // mesh.h:150-167
public:
    Plane();
    explicit Plane(const std::vector<Float3>& corners);
    RT_HOST_MOVE_ONLY(Plane)
    static std::vector<Plane> read_from_json(const std::string& class_block);
    void write_to_console(std::ostream& out) const override;
    bool intersect(const Ray& ray, Hit& hit) const override { return gpu_intersect(ray, hit); }
    using Mesh::intersect;
    void compute_aabb(Float3& outMin, Float3& outMax) const override;
private:
    int m_n_corners = 0;
};

// extension: three corners, counter-clockwise front face; no counterpart in the reference
class Triangle : public Mesh {
public:
    Triangle();
    Triangle(const Float3& a, const Float3& b, const Float3& c);
    RT_HOST_MOVE_ONLY(Triangle)
    void write_to_console(std::ostream& out) const override;
    bool intersect(const Ray& ray, Hit& hit) const override { return gpu_intersect(ray, hit); }
    using Mesh::intersect;
    void compute_aabb(Float3& outMin, Float3& outMax) const override { gpu_aabb(outMin, outMax); }
};

// mesh.h:170-198.  A node of the tree get_BVH_tree hands back.  The tree handed to the caller describes the
// flattened GPU hierarchy; traversal itself happens on the device.
class BoundingBox : public Mesh {
public:
    BoundingBox();
    BoundingBox(const Float3& bmin, const Float3& bmax);
    ~BoundingBox() override;
    BoundingBox(const BoundingBox&) = delete;
    BoundingBox& operator=(const BoundingBox&) = delete;
    BoundingBox(BoundingBox&&) noexcept = default;
    BoundingBox& operator=(BoundingBox&&) noexcept = default;

    void write_to_console(std::ostream& out) const override;
    bool intersect(const Ray& ray, Hit& hit) const override;       // slab test, mesh.cpp:942-967
    void compute_aabb(Float3& outMin, Float3& outMax) const override { outMin = m_min; outMax = m_max; }
    const BoundingBox* asBoundingBox() const override { return this; }

    void setChildren(const Mesh* left, const Mesh* right) { m_left = left; m_right = right; }
    const Mesh* getLeft() const { return m_left; }
    const Mesh* getRight() const { return m_right; }
    const Float3& getMin() const { return m_min; }
    const Float3& getMax() const { return m_max; }
    void setBounds(const Float3& bmin, const Float3& bmax) { m_min = bmin; m_max = bmax; }
private:
    Float3 m_min, m_max;
    const Mesh* m_left = nullptr;
    const Mesh* m_right = nullptr;
};

// ---- raytracer.h -------------------------------------------------------------------------------------------

struct PointLight { RayVec3 location; double radiant_intensity; };      // raytracer.h:14-17

namespace rt_host { struct GpuScene; }

// raytracer.h:20-65
class RayTracer {
public:
    RayTracer();
    ~RayTracer();
    RayTracer(const RayTracer&) = delete;
    RayTracer& operator=(const RayTracer&) = delete;

    bool load_from_json(const std::string& json_filepath);
    std::vector<Ray> get_one_ray_per_pixel(const std::string& camera_id) const;
    bool one_pass_intersection(const Ray& ray, Hit& out_hit) const;
    Pixel shade(const Hit& hit, const Ray& view_ray) const;
    bool render_unaccelerated_ppm(const std::string& camera_id, const std::string& output_filepath, int samples_per_pixel = 1,
                                  int max_scatters = 2, double min_scatter_intensity = 0.0) const;
    const std::vector<const Mesh*>& get_scene_mesh_ptrs() const { return m_scene_mesh_ptrs; }

    // extensions: whole batches in one launch; programmatic scenes; the seed of multi-sample renders (the
    // reference draws its sub-pixel offsets from std::random_device)
    std::vector<Hit> one_pass_intersection(const std::vector<Ray>& rays) const;
    std::vector<Pixel> shade(const std::vector<Hit>& hits, const std::vector<Ray>& view_rays) const;
    // brute_force: every primitive against every ray on the device, as the reference's one_pass_intersection does
    // (raytracer.cpp:143-157), instead of the walk through the hierarchy; same ids and distances either way
    bool trace_primary(const std::string& camera_id, std::vector<int>& hit_index, std::vector<double>& distance, bool brute_force = false) const;
    void set_scene(std::vector<Cube> cubes, std::vector<Plane> planes, std::vector<Cylinder> cylinders, std::vector<Sphere> spheres,
                   std::vector<Triangle> triangles, const std::vector<PointLight>& lights);
    void add_camera(const std::string& id, const Camera& cam) { m_cameras[id] = cam; }
    void set_sample_seed(unsigned seed) { m_sample_seed = seed; }
    bool render_to_image(const std::string& camera_id, Image& out, int samples_per_pixel = 1, int max_scatters = 2,
                         double min_scatter_intensity = 0.0) const;
private:
    std::vector<Cube> m_cubes;
    std::vector<Plane> m_planes;
    std::vector<Cylinder> m_cylinders;
    std::vector<Sphere> m_spheres;
    std::vector<Triangle> m_triangles;
    std::vector<const Mesh*> m_scene_mesh_ptrs;
    std::unordered_map<std::string, Camera> m_cameras;
    std::unordered_map<std::string, PointLight> m_point_lights;
    std::vector<PointLight> m_light_order;                 // file order: the order light terms are summed in
    unsigned m_sample_seed = 1;
    mutable std::unique_ptr<rt_host::GpuScene> m_gpu;      // built and uploaded on first use
    void rebuild_mesh_ptrs();
    rt_host::GpuScene& gpu() const;
    int index_of(const Mesh* m) const;
};

double time_trace_all_one_pass(const std::string& json_filepath, const std::string& camera_id);                 // raytracer.h:71

// raytracer.h:78.  Builds the GPU hierarchy over `objects` (ids = positions in `objects`), uploads it, and
// mirrors it into node_store as BoundingBox nodes; returns the root (a primitive when there is one object,
// nullptr when there are none).  The device copy lives as long as the root node does.
const Mesh* get_BVH_tree(const std::vector<const Mesh*>& objects, std::vector<std::unique_ptr<BoundingBox>>& node_store);
bool accelerated_one_pass_intersection(const Mesh* bvh_root, const Ray& ray, Hit& out_hit);                    // raytracer.h:81
std::vector<Hit> accelerated_one_pass_intersection(const Mesh* bvh_root, const std::vector<Ray>& rays);         // extension
double time_trace_all_accelerated_one_pass(const std::string& json_filepath, const std::string& camera_id);    // raytracer.h:84

#endif // RT_HOST_HPP
