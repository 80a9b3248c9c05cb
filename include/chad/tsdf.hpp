// chad/tsdf.hpp -- drop-in for the reference's public header (M2-TE/chad_tsdf include/chad/tsdf.hpp:21-171):
// the same class name, constructors, insert overloads, save() and public _sdf_res/_sdf_trunc members, backed by
// the MI355X implementation through the C ABI of chad_hip.h.  Code written against the reference header
// compiles unchanged; link libchad_hip.so instead of chad::tsdf.
//
// Differences, all outside the insert() hot path:
//   * every constructor creates the device map first (the reference's construct-and-insert constructors call
//     insert() on unallocated members, tsdf.hpp:31-44 vs tsdf.cpp:27-31);
//   * failures (no GPU, out of memory, capacity) throw std::runtime_error, as the reference does for mmap
//     failures (virtual_array.cpp:16-20); nothing is printed unless CHAD_VERBOSE=1 is set in the environment;
//   * save() reconstructs the first submap on the GPU (the reference's cell grid, lvr2.cpp:15-133, triangulated by marching
//     tetrahedra) and writes a triangle mesh as binary PLY; LVR2 itself is not used;
//   * the read-back API the reference sketches and leaves commented out (tsdf.hpp:120-161) exists: TSDFMap::iterator over a
//     closed submap's leaves (position, signed distance, weight -- no Morton codes), batched signed-distance queries and a
//     raycast on the active submap;
//   * extensions: leaves() bulk read-back, attach() for the frame-sharded multi-GPU map (chad_hip.h, "Multi-GPU").
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#if __has_include(<glm/vec3.hpp>)
#   include <glm/vec3.hpp>
#   define CHAD_HAS_GLM 1
#endif
#if __has_include(<Eigen/Eigen>)
#   include <Eigen/Eigen>
#   define CHAD_HAS_EIGEN 1
#endif

struct chad_map;   // C ABI handles (chad_hip.h)
struct chad_iter;
struct chad_comm;

namespace chad {

    class TSDFMap {
    public:
        TSDFMap(const TSDFMap&) = delete;
        TSDFMap(TSDFMap&&) = delete;
        TSDFMap& operator=(const TSDFMap&) = delete;
        TSDFMap& operator=(TSDFMap&&) = delete;

        // a map with the given voxel size and truncation distance (tsdf.hpp:29)
        TSDFMap(float sdf_res = 0.05f, float sdf_trunc = 0.1f);
        // ... then insert one scan (tsdf.hpp:31-44)
        TSDFMap(float sdf_res, float sdf_trunc, const std::vector<std::array<float, 3>>& points, const std::array<float, 3>& position)
            : TSDFMap(sdf_res, sdf_trunc) { insert(points, position); }
        TSDFMap(float sdf_res, float sdf_trunc, const float* points_p, size_t points_count, const float* position_p)
            : TSDFMap(sdf_res, sdf_trunc) { insert(points_p, points_count, position_p); }
        TSDFMap(float sdf_res, float sdf_trunc, const float* points_p, size_t points_count, float position_x, float position_y, float position_z)
            : TSDFMap(sdf_res, sdf_trunc) { insert(points_p, points_count, position_x, position_y, position_z); }
        ~TSDFMap();

        // insert a point cloud observed from the scanner position (tsdf.hpp:48)
        void insert(const std::vector<std::array<float, 3>>& points, const std::array<float, 3>& position) {
            insert(points.empty() ? nullptr : points.front().data(), points.size(), position.data());
        }
        // raw x,y,z triples; position_p points at x,y,z (tsdf.hpp:50).  No intermediate copy is made.
        void insert(const float* points_p, size_t points_count, const float* position_p);
        // raw x,y,z triples (tsdf.hpp:59)
        void insert(const float* points_p, size_t points_count, float position_x, float position_y, float position_z) {
            const float position[3] = { position_x, position_y, position_z };
            insert(points_p, points_count, position);
        }

#if defined(CHAD_HAS_GLM)
        // glm interface (tsdf.hpp:68-90)
        TSDFMap(float sdf_res, float sdf_trunc, const std::vector<glm::vec3>& points, const glm::vec3& position)
            : TSDFMap(sdf_res, sdf_trunc) { insert(points, position); }
        void insert(const std::vector<glm::vec3>& points, const glm::vec3& position) {
            if (sizeof(glm::vec3) == 12) insert(points.empty() ? nullptr : &points[0].x, points.size(), position.x, position.y, position.z);
            else {
                std::vector<std::array<float, 3>> packed;
                packed.reserve(points.size());
                for (const auto& p : points) packed.push_back({ p.x, p.y, p.z });
                insert(packed, { position.x, position.y, position.z });
            }
        }
#endif
#if defined(CHAD_HAS_EIGEN)
        // Eigen interface (tsdf.hpp:93-115)
        TSDFMap(float sdf_res, float sdf_trunc, const std::vector<Eigen::Vector3f>& points, const Eigen::Vector3f& position)
            : TSDFMap(sdf_res, sdf_trunc) { insert(points, position); }
        void insert(const std::vector<Eigen::Vector3f>& points, const Eigen::Vector3f& position) {
            std::vector<std::array<float, 3>> packed;
            packed.reserve(points.size());
            for (const auto& p : points) packed.push_back({ p.x(), p.y(), p.z() });
            insert(packed, { position.x(), position.y(), position.z() });
        }
#endif

        // tsdf.hpp:118 -- closes the active submap like the reference and writes the FIRST submap to disk
        void save(const std::string& filename);

        // ---- the read-back API of tsdf.hpp:120-161 (commented out / TODO in the reference) ----
        // what an iterator gives the user (tsdf.hpp:121-124): the leaf's real position, the data written there, no Morton codes
        struct Leaf { std::array<float, 3> position; float signed_distance; uint32_t weight; };
        // leaves of one CLOSED submap in ascending Morton order, read through the DAG (tsdf.hpp:126-151; begin() = tsdf.cpp:104-146)
        class iterator {
        public:
            iterator(const iterator&) = delete;
            iterator& operator=(const iterator&) = delete;
            iterator(iterator&& other) noexcept;
            iterator& operator=(iterator&& other) noexcept;
            ~iterator();
            bool valid() const;                 // false once moved past either end (what comparing with end() would tell)
            explicit operator bool() const { return valid(); }
            iterator& operator++();             // tsdf.hpp:137-140 (TODO in the reference)
            iterator& operator--();
            iterator& operator+=(int64_t n);
            iterator& operator-=(int64_t n);
            Leaf operator*() const;
        private:
            friend class TSDFMap;
            explicit iterator(chad_iter* it);
            chad_iter* _it = nullptr;
        };
        iterator begin(uint32_t submap) const;  // on the submap's first leaf (the reference passes the submap's root address)
        iterator last(uint32_t submap) const;   // on its last leaf (tsdf.cpp:148-153)
        // signed distance of the ACTIVE submap at arbitrary points: the containing voxel's leaf, or trilinear between the eight
        // leaves around the point; NaN where nothing was observed
        std::vector<float> signed_distance_at(const std::vector<std::array<float, 3>>& points, bool trilinear = true) const;
        // "raycast ... + physics hit" (tsdf.hpp:157): distance along each ray to the surface (sd = 0 crossing), NaN if none within max_range
        std::vector<float> raycast(const std::vector<std::array<float, 3>>& origins, const std::vector<std::array<float, 3>>& unit_directions, float max_range) const;

        // ---- extensions (not in the reference) ----
        // frame-sharded multi-GPU map: every rank calls the same inserts; see chad_hip.h, chad_map_attach_comm
        void attach(chad_comm* comm);
        struct Leaves { std::vector<uint64_t> code; std::vector<float> signed_distance; std::vector<uint32_t> weight; };
        // voxels of the active submap (submap < 0) or of a closed one, sorted by Morton code
        Leaves leaves(int submap = -1) const;
        size_t closed_submaps() const;
        // block until the device has finished all inserts
        void sync();
        chad_map* handle() const { return _impl; }

        const float _sdf_res;    // tsdf.hpp:163
        const float _sdf_trunc;  // tsdf.hpp:164
    private:
        chad_map* _impl = nullptr;
    };
}
