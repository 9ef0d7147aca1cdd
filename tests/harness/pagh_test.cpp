// tests/harness/pagh_test.cpp — TEST SUPPORT (never part of the product).
// Thin C wrapper over the product's host input pipeline so that Python tests (ctypes) can load a
// pagraph input directory and hand the SAME flat pag_build_input to both the HIP library and the oracle.
#include <cstdint>
#include <cstdio>
#include <exception>
#include <memory>
#include <string>

#include "config.hpp"
#include "graph_input.hpp"
#include "kmer_file.hpp"
#include "position_mapper.hpp"
#include <vector>

namespace {
struct Loaded {
    std::unique_ptr<pagh::KmerFile> kmers;
    std::unique_ptr<pagh::SeqDb> reads, ctgs, refs;
    std::unique_ptr<pagh::AlnDb> readToCtg, readToRef, ctgToRef;
    std::unique_ptr<pagh::GraphInput> input;   // the host restatement of the preparation stage (checker)
    pagh::BlockConfig cfg;
    pagh::BuildParams params;
    std::unique_ptr<pagh::RawInput> raw;       // what the product hands to pag_prepare
};
}  // namespace

extern "C" {

void *pagh_load(const char *dir, unsigned block, unsigned threads, uint64_t eps, uint64_t cov) {
    try {
        std::string d(dir);
        auto L = std::make_unique<Loaded>();
        auto blocks = pagh::loadConfig(d + "/config.txt");
        if (block >= blocks.size()) return nullptr;
        const auto &b = blocks[block];
        L->kmers = std::make_unique<pagh::KmerFile>(d + "/kmer.bin");
        L->ctgs = std::make_unique<pagh::SeqDb>(d + "/ctg.fasta");
        L->refs = std::make_unique<pagh::SeqDb>(d + "/ref.fasta");
        L->ctgToRef = std::make_unique<pagh::AlnDb>(d + "/aln", pagh::AlnDb::Flavor::MummerV2);
        L->reads = std::make_unique<pagh::SeqDb>(d + "/" + b.readPath);
        L->readToCtg = std::make_unique<pagh::AlnDb>(d + "/" + b.ctgAlnPath, pagh::AlnDb::Flavor::Mecat);
        L->readToRef = std::make_unique<pagh::AlnDb>(d + "/" + b.refAlnPath, pagh::AlnDb::Flavor::Mecat);
        pagh::BuildParams p;
        p.threads = threads;
        p.epsilon = eps;
        p.covFilter = cov;
        L->cfg = b;
        L->params = p;
        L->input = std::make_unique<pagh::GraphInput>(*L->reads, *L->ctgs, *L->refs, *L->readToCtg, *L->readToRef,
                                                      *L->ctgToRef, L->cfg, L->params);
        L->raw = std::make_unique<pagh::RawInput>(*L->reads, *L->ctgs, *L->refs, *L->readToCtg, *L->readToRef, *L->ctgToRef, L->cfg,
                                                  L->params);
        return L.release();
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pagh_load: %s\n", e.what());
        return nullptr;
    }
}

const pag_build_input *pagh_view(void *h) { return &static_cast<Loaded *>(h)->input->view(); }
const pag_raw_input *pagh_raw_view(void *h) { return &static_cast<Loaded *>(h)->raw->view(); }

const uint64_t *pagh_kmer_words(void *h, uint64_t *n, uint64_t *k) {
    auto *L = static_cast<Loaded *>(h);
    *n = L->kmers->size();
    *k = L->kmers->k();
    return L->kmers->data();
}

// SeqDb of a file, condensed: out[0] = sequences, out[1] = bases, out[2] = FNV-1a over names, lengths and packed bases
// (tests/test_loaders.py compares the thread-pool loaders with the sequential loops, PAGH_SEQUENTIAL_LOADERS=1)
void pagt_seqdb_digest(const char *path, uint64_t *out) {
    pagh::SeqDb db(path);
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t x) { h = (h ^ x) * 1099511628211ull; };
    for (std::size_t i = 0; i < db.size(); ++i) {
        for (char c : db.name(i)) mix(static_cast<unsigned char>(c));
        mix(0xFFu);
        mix(db.length(i));
        const std::string sq = db.toString(i, true);
        for (char c : sq) mix(static_cast<unsigned char>(c));
        mix(db.id(db.name(i)));
    }
    out[0] = db.size();
    out[1] = db.totalBases();
    out[2] = h;
}

uint64_t pagh_total_read_bases(void *h) { return static_cast<Loaded *>(h)->reads->totalBases(); }

void pagh_free(void *h) { delete static_cast<Loaded *>(h); }

// function-level hooks: the product's PositionMapper (csrc/host/position_mapper.hpp) over sequences of given lengths
static pagh::PositionMapper mapperOf(const uint32_t *len, uint64_t n) {
    pagh::SeqDb db;
    for (uint64_t i = 0; i < n; ++i) {
        std::vector<uint8_t> packed((len[i] + 3) / 4 + 16, 0);
        db.addPacked("s" + std::to_string(i), packed.data(), len[i]);
    }
    db.finish();
    return pagh::PositionMapper(db);
}
uint64_t pagt_mapper_d2s(const uint32_t *len, uint64_t n, int64_t idx, int64_t pos) { return mapperOf(len, n).dualToSingle(idx, pos); }
void pagt_mapper_s2d(const uint32_t *len, uint64_t n, uint64_t single, int64_t *idx, int64_t *pos) {
    auto d = mapperOf(len, n).singleToDual(single);
    *idx = d.first;
    *pos = d.second;
}
uint64_t pagt_mapper_extra(const uint32_t *len, uint64_t n) { return mapperOf(len, n).extraStart(); }

}  // extern "C"

// test hook for SeqDb's pack window (seq_db.hpp): a FASTQ file loaded whole and as rank `rank` of `world` (emission order over
// `threads` workers).  out[0] = records, out[1] = records inside the window, out[2] = 1 if names and lengths agree for every
// record, every record inside the window reads back base for base, every other one points at the shared zero stretch;
// out[3] / out[4] = packed bytes of the windowed / the whole load.
extern "C" int pagt_seq_window_check(const char *path, unsigned rank, unsigned world, unsigned threads, uint64_t *out) {
    try {
        pagh::SeqDb full(path);
        const pagh::SeqDb::PackWindow w{rank, world, threads};
        pagh::SeqDb part(path, &w);
        out[0] = full.size();
        out[1] = 0;
        bool ok = full.size() == part.size() && full.totalBases() == part.totalBases();
        std::uint64_t sharedAt = ~0ull;
        for (std::size_t i = 0; ok && i < full.size(); ++i) {
            ok = full.name(i) == part.name(i) && full.length(i) == part.length(i) && part.id(part.name(i)) == full.id(full.name(i));
            if (w.wants(i, full.size())) {
                out[1] += 1;
                ok = ok && full.toString(i, true) == part.toString(i, true) && full.toString(i, false) == part.toString(i, false);
            } else {
                if (sharedAt == ~0ull) sharedAt = part.byteOff()[i];
                ok = ok && part.byteOff()[i] == sharedAt && part.toString(i, true) == std::string(part.length(i), 'A');
            }
        }
        out[2] = ok ? 1 : 0;
        out[3] = part.packed().size();
        out[4] = full.packed().size();
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pagt_seq_window_check: %s\n", e.what());
        return 1;
    }
}

// test hook for AlnDb's record filter (aln_db.hpp, AlnRecordFilter): the file parsed without a filter and with "numeric query
// name % mod == rem".  out[0] = records, out[1] = records that kept their columns, out[2] = 1 if every header field of every
// record agrees, every kept record has the unfiltered parse's column classes and counts, and every other record has none.
extern "C" int pagt_aln_filter_check(const char *path, uint64_t mod, uint64_t rem, uint64_t *out) {
    try {
        pagh::AlnDb full(path, pagh::AlnDb::Flavor::Mecat);
        pagh::AlnDb part(path, pagh::AlnDb::Flavor::Mecat, [mod, rem](const char *name, std::size_t len) { return std::stoull(std::string(name, len)) % mod == rem; });
        out[0] = full.size();
        out[1] = 0;
        bool ok = full.size() == part.size();
        for (std::size_t i = 0; ok && i < full.size(); ++i) {
            const pagh::AlnRecord &a = full[i], &b = part[i];
            ok = a.queryName == b.queryName && a.refName == b.refName && a.score == b.score && a.queryBegin == b.queryBegin && a.queryEnd == b.queryEnd &&
                 a.refBegin == b.refBegin && a.refEnd == b.refEnd && a.forward == b.forward;
            const bool mine = std::stoull(a.queryName) % mod == rem;
            if (mine) {
                out[1] += 1;
                ok = ok && a.nCols == b.nCols && a.nEmit == b.nEmit && a.nRadv == b.nRadv;
                for (std::size_t c = 0; ok && c < a.nCols; ++c) ok = full.colClass(a, c) == part.colClass(b, c);
            } else {
                ok = ok && b.nCols == 0 && b.nEmit == 0 && b.nRadv == 0;
            }
        }
        out[2] = ok ? 1 : 0;
        out[3] = part.diff().size();
        out[4] = full.diff().size();
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pagt_aln_filter_check: %s\n", e.what());
        return 1;
    }
}
