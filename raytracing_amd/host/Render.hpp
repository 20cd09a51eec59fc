/* Render.hpp -- ray_trace_from_camera() of R/main.cpp:170-221 for a headless host: the
 * per-pixel / per-sample loop runs on the GPU (rt_render), the host keeps the timing print and
 * the ASCII P3 writer with the reference's quantisation (int)(255.f * min(1.f, c)). */
#pragma once
#include <algorithm>
#include <chrono>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include "Camera.hpp"
#include "Scene.hpp"

inline bool write_ppm_p3(const std::string& filename, const std::vector<Vec3>& image, int w, int h) {
    std::ofstream f(filename.c_str(), std::ios::binary);
    if (f.fail()) {
        std::cout << "Could not open file: " << filename << std::endl;
        return false;
    }
    f << "P3" << std::endl << w << " " << h << std::endl << 255 << std::endl;
    for (int i = 0; i < w * h; ++i)
        f << (int)(255.f * std::min<float>(1.f, image[i][0])) << " " << (int)(255.f * std::min<float>(1.f, image[i][1])) << " "
          << (int)(255.f * std::min<float>(1.f, image[i][2])) << " ";
    f << std::endl;
    return true;
}

/* the same pixels as binary P6 (one sixth of the bytes, no text parsing for the reader) */
inline bool write_ppm_p6(const std::string& filename, const std::vector<Vec3>& image, int w, int h) {
    std::vector<uint8_t> rgb8((size_t)w * h * 3);
    for (int i = 0; i < w * h; ++i)
        for (int c = 0; c < 3; ++c) rgb8[3 * (size_t)i + c] = (uint8_t)(int)(255.f * std::min<float>(1.f, image[i][c]));
    if (rt_write_ppm(filename.c_str(), rgb8.data(), (uint32_t)w, (uint32_t)h, 1) != 0) {
        std::cout << "Could not open file: " << filename << std::endl;
        return false;
    }
    return true;
}

inline void ray_trace_from_camera(Scene& scene, Camera& camera, unsigned nsamples = 50, uint32_t seed = 1,
                                  const std::string& filename = "./rendu.ppm", rt_stats* stats_out = nullptr,
                                  const std::string& filename_p6 = "") {
    int w = camera.getScreenWidth(), h = camera.getScreenHeight();
    std::cout << "Ray tracing a " << w << " x " << h << " image" << std::endl;
    auto start = std::chrono::high_resolution_clock::now();
    camera.apply();
    std::vector<Vec3> image;
    rt_stats st{};
    scene.render(camera.glState(), w, h, nsamples, seed, image, &st);
    if (stats_out) *stats_out = st;
    if (rt_comm_rank() != 0) return; /* several processes, one per GPU: rank 0 has the image and writes it */
    std::cout << "\rProgress: 100% - Done!            " << std::endl;
    if (!write_ppm_p3(filename, image, w, h)) return;
    if (!filename_p6.empty() && !write_ppm_p6(filename_p6, image, w, h)) return;
    auto end = std::chrono::high_resolution_clock::now();
    auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count();
    std::cout << "Image saved to '" << filename << "' in " << ms << " ms." << std::endl;
    std::cout << st.rays << " rays, kernel " << st.kernel_ms << " ms (" << (st.kernel_ms > 0 ? st.rays / st.kernel_ms / 1e3 : 0.0) << " Mrays/s)" << std::endl;
    rt_sharded_times tm{};
    if (rt_sharded_last_times(&tm) == RT_OK)
        std::cout << "frame on rank 0 of " << rt_comm_world() << ": render " << tm.render_ms << " ms, gather " << tm.gather_ms << " ms, un-tile " << tm.untile_ms
                  << " ms, copy to host " << tm.copy_ms << " + " << tm.host_copy_ms << " ms, call " << tm.total_ms << " ms" << std::endl;
    if (stats_out) *stats_out = st;
}
