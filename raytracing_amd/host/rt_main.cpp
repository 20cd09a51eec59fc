/* rt_main.cpp -- headless stand-in for the GLUT shell of R/main.cpp:224-357.
 *
 * The window's two actions that matter for rendering are reachable from the command line:
 *   '+'  select the next scene          (R/main.cpp:255-258)
 *   'r'  render the selected scene      (R/main.cpp:250-254)
 * Usage: rt_main [--w 480] [--h 480] [--spp 50] [--seed 1] [--out ./rendu.ppm] [--p6 file] [--off img/star.off] [--keys r]
 *        (--p6: also write the image as binary P6)
 * e.g.  rt_main --keys +r   renders scene 3 (Cornell box + mesh) like the committed rendu.ppm. */
#include <cstring>
#include <string>
#include "Render.hpp"

int main(int argc, char** argv) {
    unsigned w = 480, h = 480, spp = 50, seed = 1; /* R/main.cpp:46-47, :181 */
    std::string out = "./rendu.ppm", keys = "r", off = "img/star.off", p6;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i], v = argv[i + 1];
        if (k == "--w") w = std::stoul(v);
        else if (k == "--h") h = std::stoul(v);
        else if (k == "--spp") spp = std::stoul(v);
        else if (k == "--seed") seed = std::stoul(v);
        else if (k == "--out") out = v;
        else if (k == "--keys") keys = v;
        else if (k == "--p6") p6 = v;
        else if (k == "--off") off = v;
        else { std::cerr << "unknown option " << k << std::endl; return EXIT_FAILURE; }
    }
    Camera camera;
    camera.resize(w, h);
    camera.move(0., 0., -3.1);
    unsigned selected_scene = 2;
    std::vector<Scene> scenes(4);
    scenes[0].setup_single_sphere();
    scenes[1].setup_single_square();
    scenes[2].setup_cornell_box();
    scenes[3].setup_cornell_box_meshes(off);
    for (auto& mesh : scenes[selected_scene].getMeshes()) mesh.computeAABB();
    scenes[selected_scene].buildKDTree();
    for (char key : keys) {
        if (key == '+') { if (++selected_scene >= scenes.size()) selected_scene = 0; }
        else if (key == 'r') { camera.apply(); ray_trace_from_camera(scenes[selected_scene], camera, spp, seed, out, nullptr, p6); }
    }
    rt_shutdown();
    return EXIT_SUCCESS;
}
