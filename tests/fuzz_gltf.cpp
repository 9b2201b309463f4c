// ASAN/UBSAN harness for the header-only glTF loader: loads every file named on the command line.
#include <cstdio>
#include <string>
#include "host/gltf_loader.hpp"
using namespace raytracer;
int main(int argc, char** argv) {
    int ok = 0, bad = 0;
    for (int i = 1; i < argc; i++) {
        SceneState s;
        GltfError e = scene_state_load_from_gltf(argv[i], s);
        if (e) bad++; else ok++;
    }
    std::printf("loaded %d, rejected %d\n", ok, bad);
    return 0;
}
