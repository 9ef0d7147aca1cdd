// pagraph — MI355X-native drop-in for AlignGraph2's PAGraph stage (reference PAGraph/src/main/pagraph.cpp).
// Same command line, same input files, same output files; the graph build runs on the GPU through
// libpagraph_hip.so.  There is no CPU build path in this program.
#include <cstdlib>

#include "hip_backend.hpp"

int main(int argc, char **argv) {
    const char *dev = std::getenv("PAGRAPH_DEVICE");
    auto backend = pagh::makeHipBackend(dev ? std::atoi(dev) : 0);
    return pagh::runPagraph(argc, argv, *backend);
}
