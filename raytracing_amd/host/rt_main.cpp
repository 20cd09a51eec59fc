/* rt_main.cpp -- headless stand-in for the GLUT shell of R/main.cpp:224-357.
 *
 * The window's two actions that matter for rendering are reachable from the command line:
 *   '+'  select the next scene          (R/main.cpp:255-258)
 *   'r'  render the selected scene      (R/main.cpp:250-254)
 * Usage: rt_main [--w 480] [--h 480] [--spp 50] [--seed 1] [--scene 2] [--gpus 1] [--out ./rendu.ppm] [--p6 file]
 *                [--off img/star.off] [--keys r]
 *        --scene: the scene selected at start (the reference starts with 2, R/main.cpp:347); --p6: also write the image as
 *        binary P6; --gpus N: N processes, one per GPU (devices 0 .. N-1), each renders the tiles t % N == its rank and
 *        rank 0 gathers them over RCCL and writes the file.
 * e.g.  rt_main --keys +r   renders scene 3 (Cornell box + mesh) like the committed rendu.ppm. */
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <string>
#include "Render.hpp"

/* The processes of a --gpus run agree on a communicator through a file: rank 0 writes the id, the others wait for it. */
static bool join_communicator(int rank, int world, const std::string& id_file) {
    char id[RT_COMM_ID_BYTES];
    if (rank == 0) {
        rthost::require(rt_comm_unique_id(id), "rt_comm_unique_id");
        const std::string tmp = id_file + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id) { std::cerr << "could not write " << tmp << std::endl; return false; }
        std::fclose(f);
        if (std::rename(tmp.c_str(), id_file.c_str()) != 0) { std::cerr << "could not publish " << id_file << std::endl; return false; }
    } else {
        bool got = false;
        for (int tries = 0; tries < 600 && !got; ++tries) { /* up to a minute */
            FILE* f = std::fopen(id_file.c_str(), "rb");
            if (f) { got = std::fread(id, 1, sizeof id, f) == sizeof id; std::fclose(f); }
            if (!got) usleep(100000);
        }
        if (!got) { std::cerr << "rank " << rank << ": no communicator id in " << id_file << std::endl; return false; }
    }
    rthost::require(rt_comm_init(id, rank, world), "rt_comm_init");
    return true;
}

int main(int argc, char** argv) {
    unsigned w = 480, h = 480, spp = 50, seed = 1; /* R/main.cpp:46-47, :181 */
    unsigned selected_scene = 2, gpus = 1;         /* R/main.cpp:347 */
    std::string out = "./rendu.ppm", keys = "r", off = "img/star.off", p6;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i], v = argv[i + 1];
        if (k == "--w") w = std::stoul(v);
        else if (k == "--h") h = std::stoul(v);
        else if (k == "--spp") spp = std::stoul(v);
        else if (k == "--seed") seed = std::stoul(v);
        else if (k == "--scene") selected_scene = std::stoul(v);
        else if (k == "--gpus") gpus = std::stoul(v);
        else if (k == "--out") out = v;
        else if (k == "--keys") keys = v;
        else if (k == "--p6") p6 = v;
        else if (k == "--off") off = v;
        else { std::cerr << "unknown option " << k << std::endl; return EXIT_FAILURE; }
    }
    if (selected_scene > 3 || gpus < 1 || gpus > 64) { std::cerr << "--scene must be 0..3, --gpus 1..64" << std::endl; return EXIT_FAILURE; }
    int rank = 0;
    if (gpus > 1) {
        /* one process per GPU, forked before anything has touched a GPU; the parent only waits for them */
        const std::string id_file = "/tmp/rtamd_comm_" + std::to_string((long)getpid());
        std::remove(id_file.c_str()); /* a file left by an earlier run with this process id would be read as a valid id */
        std::vector<pid_t> kids;
        bool child = false;
        for (unsigned r = 0; r < gpus && !child; ++r) {
            pid_t pid = fork();
            if (pid < 0) { std::perror("fork"); return EXIT_FAILURE; }
            if (pid == 0) { child = true; rank = (int)r; }
            else kids.push_back(pid);
        }
        if (!child) {
            /* the first rank that fails ends the run: its peers may already wait for it inside the communicator's set-up or
             * an exchange, where nothing would ever wake them */
            int worst = EXIT_SUCCESS;
            size_t left = kids.size();
            while (left > 0) {
                int status = 0;
                const pid_t pid = waitpid(-1, &status, 0);
                if (pid < 0) { worst = EXIT_FAILURE; break; }
                --left;
                for (pid_t& k : kids) if (k == pid) k = 0;
                if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) {
                    worst = EXIT_FAILURE;
                    for (pid_t k : kids) if (k > 0) kill(k, SIGTERM);
                }
            }
            std::remove(id_file.c_str());
            return worst;
        }
        int n_devices = 0;
        if (rt_device_count(&n_devices) != RT_OK || (unsigned)n_devices < gpus) {
            std::cerr << "rank " << rank << ": --gpus " << gpus << " but " << n_devices << " HIP device(s) visible" << std::endl;
            return EXIT_FAILURE; /* before anybody waits for this rank */
        }
        setenv("RT_DEVICE", std::to_string(rank).c_str(), 1);
        rthost::ensure_device();
        if (!join_communicator(rank, (int)gpus, id_file)) return EXIT_FAILURE;
    }
    Camera camera;
    camera.resize(w, h);
    camera.move(0., 0., -3.1);
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
    rt_comm_destroy();
    rt_shutdown();
    return EXIT_SUCCESS;
}
