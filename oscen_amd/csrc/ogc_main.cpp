// ogc -- ahead-of-time front end of the graph compiler: prints the generated
// HIP translation unit of a built-in graph.  `python -m oscen_amd.build` runs
// it for every built-in graph and compiles the result with hipcc into
// liboscen_gpu.so (csrc/gen/<name>.hip is committed so the kernels are
// reviewable; the build re-generates and checks them).
#include <cstdio>
#include <cstring>
#include <iostream>

#include "og_graph.h"

int main(int argc, char** argv)
{
    try {
        if (argc == 2 && !strcmp(argv[1], "--list")) {
            for (auto& n : ogc::builtin_graph_names()) std::cout << n << "\n";
            return 0;
        }
        if (argc != 2) {
            fprintf(stderr, "usage: ogc <builtin-graph-name> | --list\n");
            return 2;
        }
        auto cg = ogc::compile(ogc::builtin_graph(argv[1]));
        std::cout << cg->source;
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "ogc: %s\n", e.what());
        return 1;
    }
}
