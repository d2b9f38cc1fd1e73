// Exercises include/yuki_hip.hpp: host-side pieces always; with a GPU (argv[1] == "gpu")
// Integrator::render on one tile and render_tiles on the whole film.  Prints
// key=value lines that tests/test_cxx_mirror.py checks.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "yuki_hip.hpp"

int main(int argc, char** argv) {
    using namespace yuki;
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    // two stacked quads (4 triangles) lit by a point light
    std::vector<float> pts = {-1, 0, -1, 1, 0, -1, 1, 0, 1, -1, 0, 1, -0.5f, 0.5f, -0.5f, 0.5f, 0.5f, -0.5f, 0.5f, 0.5f, 0.5f, -0.5f, 0.5f, 0.5f};
    std::vector<uint32_t> idx = {0, 2, 1, 0, 3, 2, 4, 6, 5, 4, 7, 6};
    std::vector<uint32_t> tri_mesh = {0, 0, 0, 0};
    std::vector<int32_t> tri_mat = {0, 0, 1, 1}, tri_al = {-1, -1, -1, -1};
    yk_mesh_desc mesh = {0, 0, 0, 0};
    yk_material_desc mats[2] = {};
    mats[0].kind = YK_MAT_MATTE;
    mats[0].a[0] = mats[0].a[1] = mats[0].a[2] = 0.8f;
    mats[1].kind = YK_MAT_GLASS;
    for (int k = 0; k < 3; ++k) mats[1].a[k] = mats[1].b[k] = 1.0f;
    mats[1].c = 1.5f;
    const float l2w[16] = {1, 0, 0, 0, 0, 1, 0, 3, 0, 0, 1, 0, 0, 0, 0, 1};
    const float intensity[3] = {30, 30, 30};
    yk_light_desc light;
    check(yk_make_point_light(l2w, intensity, &light));
    yk_scene_desc d{};
    d.n_vertices = 8;
    d.points = pts.data();
    d.n_triangles = 4;
    d.indices = idx.data();
    d.tri_mesh = tri_mesh.data();
    d.tri_material = tri_mat.data();
    d.tri_area_light = tri_al.data();
    d.n_meshes = 1;
    d.meshes = &mesh;
    d.n_materials = 2;
    d.materials = mats;
    d.n_lights = 1;
    d.lights = &light;
    d.background[0] = d.background[1] = d.background[2] = 0.1f;
    d.split_method = YK_SPLIT_SAH;
    d.max_shapes_in_node = 1;

    FilmSettings fs;
    fs.res_x = 40;
    fs.res_y = 24;
    std::vector<FilmTile> tiles = film_tiles(fs);
    std::printf("tiles=%zu first=%u,%u last=%u,%u\n", tiles.size(), tiles.front().x0, tiles.front().y0, tiles.back().x1, tiles.back().y1);
    CameraParameters cp;
    cp.position = {0, 2, -3};
    cp.target = {0, 0.3f, 0};
    cp.fov_degrees = 50;
    Camera cam(cp, fs);
    std::printf("camera_c2w_03=%.6f\n", cam.matrices.camera_to_world[3]);
    {
        Scene host(nullptr, d);
        auto bvh = host.export_bvh();
        std::printf("nodes=%zu shapes=%zu depth=%u\n", bvh.first.size(), bvh.second.size(), host.info().tree_depth);
    }
    try {
        d.max_shapes_in_node = 0;
        Scene bad(nullptr, d);
        std::printf("bad_scene=accepted\n");
    } catch (const Error& e) {
        std::printf("bad_scene=status%d\n", (int)e.status);
    }
    d.max_shapes_in_node = 1;
    // scene::pbrt::load through the mirror (argv[2] = a .pbrt file written by the test)
    if (argc > 2) {
        LoadedScene ls(argv[2], LoadedScene::Format::Pbrt);
        Scene loaded(nullptr, ls.desc);
        std::printf("pbrt_triangles=%u pbrt_spheres=%u pbrt_lights=%u pbrt_res=%ux%u pbrt_fov=%.1f pbrt_nodes=%llu\n", ls.desc.n_triangles, ls.desc.n_spheres, ls.desc.n_lights,
                    ls.film.res_x, ls.film.res_y, ls.camera.fov_degrees, (unsigned long long)loaded.info().n_nodes);
        try {
            LoadedScene missing(std::string(argv[2]) + ".nope", LoadedScene::Format::Pbrt);
            std::printf("missing_scene=accepted\n");
        } catch (const Error& e) {
            std::printf("missing_scene=status%d\n", (int)e.status);
        }
    }
    if (!gpu) return 0;

    Context ctx(0);
    Scene scene(&ctx, d);
    Integrator path(ctx, IntegratorType::Path(PathParams{6, false, 0.0f}));
    yk_sampler_desc smp = SamplerType::Stratified(2, 2);
    std::vector<float> film((size_t)fs.res_x * fs.res_y * 3);
    yk_render_stats st = path.render_tiles(scene, cam, smp, tiles, film.data());
    double sum = 0;
    for (float v : film) sum += v;
    std::printf("rays=%llu samples=%llu mean=%.6f\n", (unsigned long long)st.rays, (unsigned long long)st.samples, sum / film.size());
    // the trait method on the first tile equals the first tile of the batch
    std::vector<float> tile_px(16 * 16 * 3);
    size_t rays = path.render(scene, cam, smp, tiles[0], tile_px.data());
    int w = tiles[0].x1 - tiles[0].x0, h = tiles[0].y1 - tiles[0].y0;
    bool same = std::memcmp(tile_px.data(), film.data(), (size_t)w * h * 12) == 0;
    std::printf("tile_rays=%zu tile_matches_batch=%d\n", rays, same ? 1 : 0);
    // accumulating film: the sum of the four sample passes / 4 equals the plain render
    {
        std::vector<float> acc(film.size(), 0.0f), pass(film.size());
        for (uint16_t sidx = 0; sidx < 4; ++sidx) {
            path.render_tiles_accumulating(scene, cam, smp, tiles, std::vector<uint16_t>(tiles.size(), sidx), pass.data());
            for (size_t k = 0; k < acc.size(); ++k) acc[k] += pass[k];
        }
        bool eq = true;
        for (size_t k = 0; k < acc.size(); ++k) eq = eq && (acc[k] / 4.0f == film[k]);
        std::printf("accumulate_matches_plain=%d\n", eq ? 1 : 0);
    }
    // the reference's render workers as they are (render_manager.rs:78-97): six threads, each rendering one tile at a time through the
    // combiner; every tile equals its slab of the batched film and the ray counts add up
    {
        Context lane2(0);
        Combiner comb({&ctx, &lane2}, 0, 200);
        std::vector<float> out(film.size(), -1.0f);
        std::vector<size_t> offs(tiles.size() + 1, 0);
        for (size_t t = 0; t < tiles.size(); ++t) offs[t + 1] = offs[t] + (size_t)(tiles[t].x1 - tiles[t].x0) * (tiles[t].y1 - tiles[t].y0);
        std::atomic<size_t> next{0}, total_rays{0};
        std::vector<std::thread> workers;
        for (int k = 0; k < 6; ++k)
            workers.emplace_back([&] {
                for (size_t t = next++; t < tiles.size(); t = next++)
                    total_rays += comb.render(scene, cam, smp, IntegratorType::Path(PathParams{6, false, 0.0f}), tiles[t], out.data() + offs[t] * 3);
            });
        for (auto& w : workers) w.join();
        yk_combiner_info ci = comb.info();
        std::printf("combiner_matches_batch=%d combiner_rays_match=%d combiner_merged=%d\n", std::memcmp(out.data(), film.data(), film.size() * 4) == 0 ? 1 : 0,
                    total_rays.load() == st.rays ? 1 : 0, (ci.tiles == tiles.size() && ci.submissions < ci.tiles) ? 1 : 0);
    }
    // the GPUs of the process behind one object (here: device 0, the RCCL exchange looped back to itself): the film
    // equals Film::update_tile of the batched render above
    {
        Node node({0});
        node.set_option("rccl_loopback", 1);
        node.set_scene(d);
        node.set_film(fs);
        std::vector<float> whole((size_t)fs.res_x * fs.res_y * 3), ref(whole.size());
        yk_render_stats ms = node.render_film(cam, smp, IntegratorType::Path(PathParams{6, false, 0.0f}), whole.data());
        yk_film_update_tiles(tiles.data(), tiles.size(), film.data(), fs.res_x, fs.res_y, ref.data());
        std::printf("node_rays_match=%d node_film_matches=%d node_devices=%u\n", ms.rays == st.rays ? 1 : 0, std::memcmp(whole.data(), ref.data(), whole.size() * 4) == 0 ? 1 : 0,
                    node.device_count());
        try {
            Node two({0, 0});
            std::printf("node_dup=accepted\n");
        } catch (const Error& e) {
            std::printf("node_dup=status%d\n", (int)e.status);
        }
        // three virtual ranks on device 0 (YK_MULTI_SHARED_DEVICES): the same film; then the accumulating film, 2 + 2 passes = 4 x the plain one
        Node three({0, 0, 0}, YK_MULTI_SHARED_DEVICES);
        three.set_scene(d);
        three.set_film(fs);
        std::vector<float> v3(whole.size());
        yk_render_stats m3 = three.render_film(cam, smp, IntegratorType::Path(PathParams{6, false, 0.0f}), v3.data());
        std::printf("virtual_ranks_match=%d\n", (m3.rays == st.rays && std::memcmp(v3.data(), ref.data(), v3.size() * 4) == 0) ? 1 : 0);
        three.clear_film();
        three.accumulate_film(cam, smp, IntegratorType::Path(PathParams{6, false, 0.0f}), 0, 2, nullptr);
        three.accumulate_film(cam, smp, IntegratorType::Path(PathParams{6, false, 0.0f}), 2, 2, v3.data());
        bool acc_ok = true;
        for (size_t k = 0; k < v3.size(); ++k) acc_ok = acc_ok && (v3[k] / 4.0f == ref[k]);
        std::printf("virtual_ranks_accumulate=%d\n", acc_ok ? 1 : 0);
        uint64_t px = 0, total = 0;
        size_t n_tiles = 0;
        for (uint32_t r = 0; r < 3; ++r) {
            n_tiles += Node::deal(fs, 3, r, &px).size();
            total += px;
        }
        std::printf("deal_covers_film=%d\n", (n_tiles == tiles.size() && total == (uint64_t)fs.res_x * fs.res_y) ? 1 : 0);
    }
    try {
        FilmTile bad{8, 8, 8, 12};
        path.render(scene, cam, smp, bad, tile_px.data());
        std::printf("bad_tile=accepted\n");
    } catch (const Error& e) {
        std::printf("bad_tile=status%d\n", (int)e.status);
    }
    return 0;
}
