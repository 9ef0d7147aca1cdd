// tests/harness/oracle_graph_dump.cpp — TEST HARNESS (never part of the product).
//
// Runs the product's host-side input pipeline (parsers + GraphInput) and feeds the flat input to the
// C ORACLE (oracle/pag_oracle.c), then dumps the graph in the same text format as
// oracle/ref_harness/graph_dump.cpp, so that  reference graph == oracle graph  can be checked with a
// byte comparison (tests/test_oracle_golden.py).  This pins the oracle AND the host input pipeline.
//
// usage: oracle_graph_dump -t T -k kmer.bin -c ctg.fasta -R ref.fasta -p predir -a aln -o outdir [--epsilon E] [-v V]
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "config.hpp"
#include "graph_input.hpp"
#include "kmer_file.hpp"
#include "pag_oracle.h"

static const char *argOf(int argc, char **argv, const char *flag, const char *dflt) {
    for (int i = 1; i + 1 < argc; ++i)
        if (std::strcmp(argv[i], flag) == 0) return argv[i + 1];
    return dflt;
}

int main(int argc, char **argv) {
    using namespace pagh;
    BuildParams p;
    p.threads = static_cast<unsigned>(std::atoi(argOf(argc, argv, "-t", "1")));
    p.epsilon = static_cast<std::size_t>(std::atoll(argOf(argc, argv, "--epsilon", "10")));
    p.covFilter = static_cast<std::size_t>(std::atoll(argOf(argc, argv, "-v", "1")));
    std::string preDir = argOf(argc, argv, "-p", "");
    std::string outDir = argOf(argc, argv, "-o", ".");

    KmerFile kf(argOf(argc, argv, "-k", ""));
    SeqDb ctgs(argOf(argc, argv, "-c", ""));
    SeqDb refs(argOf(argc, argv, "-R", ""));
    AlnDb ctgToRef(argOf(argc, argv, "-a", ""), AlnDb::Flavor::MummerV2);
    pago_graph *g = pago_create(kf.data(), kf.size(), static_cast<uint32_t>(kf.k()));

    auto blocks = loadConfig(preDir + "/config.txt");
    std::size_t blockNo = 0;
    for (auto &b : blocks) {
        pago_reset(g);
        SeqDb reads(preDir + "/" + b.readPath);
        AlnDb readToCtg(preDir + "/" + b.ctgAlnPath, AlnDb::Flavor::Mecat);
        AlnDb readToRef(preDir + "/" + b.refAlnPath, AlnDb::Flavor::Mecat);
        GraphInput gi(reads, ctgs, refs, readToCtg, readToRef, ctgToRef, b, p);
        pag_build_stats st;
        if (pago_process(g, &gi.view(), &st) != PAG_OK) return 2;

        uint64_t nn, np, ne;
        pago_csr_sizes(g, &nn, &np, &ne);
        std::vector<uint32_t> code(nn), pc(np), pr(np), et(ne);
        std::vector<uint16_t> cnt(np);
        std::vector<int32_t> es(ne);
        std::vector<uint64_t> poff(nn + 1), eoff(nn + 1);
        pag_csr csr{nn, np, ne, code.data(), poff.data(), pc.data(), pr.data(), cnt.data(), eoff.data(), et.data(), es.data()};
        if (pago_export_csr(g, &csr) != PAG_OK) return 3;

        std::ofstream out(outDir + "/" + std::to_string(blockNo) + ".graph.txt");
        out << "S " << st.merge_edge[0] << " " << st.total_pos[0] << " " << st.merge_pos[0] << " " << st.merge_edge[1]
            << " " << st.total_pos[1] << " " << st.merge_pos[1] << "\n";
        for (uint64_t i = 0; i < nn; ++i) {
            out << "K " << code[i] << " " << (poff[i + 1] - poff[i]) << " " << (eoff[i + 1] - eoff[i]) << "\n";
            for (uint64_t j = poff[i]; j < poff[i + 1]; ++j) out << "P " << pc[j] << " " << pr[j] << " " << cnt[j] << "\n";
            for (uint64_t j = eoff[i]; j < eoff[i + 1]; ++j) out << "C " << et[j] << " " << es[j] << "\n";
        }
        ++blockNo;
    }
    pago_destroy(g);
    return 0;
}
