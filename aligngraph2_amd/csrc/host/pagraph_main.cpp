// pagraph — MI355X-native drop-in for AlignGraph2's PAGraph stage (reference PAGraph/src/main/pagraph.cpp).
// Same command line, same input files, same output files; the graph build runs on the GPU through
// libpagraph_hip.so.  There is no CPU build path in this program.
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <iostream>
#include <memory>

#include "hip_backend.hpp"

int main(int argc, char **argv) {
    const char *dev = std::getenv("PAGRAPH_DEVICE");
    std::unique_ptr<pagh::GraphBackend> backend;
    try {
        backend = pagh::makeHipBackend(dev ? std::atoi(dev) : 0);
    } catch (const std::exception &e) {  // (a bad PAGRAPH_SHARD / PAGRAPH_SHARD_DIR, a communicator that cannot be set up)
        std::cerr << "pagraph: " << e.what() << std::endl;
        return 1;
    }
    const int rc = pagh::runPagraph(argc, argv, *backend);
    // Every output file is written and closed, every host thread joined: the process leaves without returning tens of GB of
    // device and pinned memory piece by piece (the driver reclaims them in one go: 0.1-0.2 s of a 2 s run).
    // (a rank of a sharded build leaves the ordinary way: its communicator says goodbye to its peers)
    const char *shard = std::getenv("PAGRAPH_SHARD");
    // (tools that flush at exit — rocprofv3, coverage, sanitizers — need the ordinary way out as well)
    bool tool = false;
    for (const char *name : {"ROCPROFILER_REGISTER_FORCE_LOAD", "ROCP_TOOL_LIBRARIES", "LD_PRELOAD", "GCOV_PREFIX", "ASAN_OPTIONS", "LSAN_OPTIONS"})
        tool = tool || std::getenv(name) != nullptr;
    if (!(shard && *shard) && !tool) {
        std::cout.flush();
        std::cerr.flush();
        std::fflush(nullptr);
        std::_Exit(rc);
    }
    return rc;
}
