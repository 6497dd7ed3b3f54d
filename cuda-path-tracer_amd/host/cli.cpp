// nori_hip_cli scene.xml — headless stand-in for the reference's `nori scene.xml` (src/main.cpp:23-59).
#include "render.h"
#include <iostream>
int main(int argc, char **argv) {
    if (argc != 2) { std::cerr << "Syntax: " << argv[0] << " <scene.xml>" << std::endl; return -1; }
    try {
        std::string out = nori::renderScene(argv[1]);
        if (!out.empty()) std::cout << "Wrote " << out << std::endl;
    } catch (const std::exception &e) { std::cerr << "Fatal error: " << e.what() << std::endl; return -1; }
    return 0;
}
