// k_ingest.hip — a block's TEXT inputs to the packed forms the build works on, on the device (SURVEY.md 8f.2 first half;
// north_star's "2-bit packing from aligned read blocks"):
//
//   pag_pack_text_seqs     CompressedSeq::CompressedSeq (PAGraph/src/tools/seq/CompressedSeq.cpp:8-38): the bases of a FASTA /
//                          FASTQ record, 4 per byte, base i at bits 2 * (i & 3); C/c = 1, G/g = 2, T/t = 3, every other
//                          character 0 (SeqHelper feeds the record's sequence lines as they are).
//   pag_classify_columns   parseDiff (PAGraph/src/tools/align/ParseAlignTools.cpp:8-26) in the form the alignment database
//                          keeps it (aln_db.cpp): per column of the query row 2 bits — 1: the query has a gap, 2: the
//                          reference row has one, 3: the characters differ (a reference row shorter than the query row reads
//                          as NUL there), 0: they are equal — 16 columns per u32, and per record the columns that emit a
//                          query base (class != 1) and those that advance the reference (class != 2).
//
// The host keeps what is not bulk: finding the lines (memchr), names and the header fields of the ALN records (stream
// extraction semantics, aln_db.cpp).  The text comes in as it lies in the file — resident in HBM, or in host memory, from
// where it is streamed through two pinned staging buffers (copies by a few host threads, the upload of one chunk under
// the copy of the next).  Both kernels are streaming: one read of the text, 0.25 B / base or column written.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "pag_device.hpp"

namespace pagdev {
namespace {

__device__ __forceinline__ uint32_t base_code(uint32_t c) {
    c &= 0xDFu;  // (upper case: 'a' ^ 'A' = 0x20; the four letters are unaffected by the bit in their upper-case form)
    return c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u;
}

// 16 consecutive bytes from p + i, zero past `n`; p itself may have any alignment
__device__ __forceinline__ void load16(const unsigned char *__restrict__ p, uint64_t i, uint64_t n, uint32_t out[4]) {
    const uintptr_t a = (uintptr_t)(p + i);
    if ((a & 3u) == 0 && i + 16 <= n) {
        const uint32_t *q = (const uint32_t *)(p + i);
        out[0] = q[0];
        out[1] = q[1];
        out[2] = q[2];
        out[3] = q[3];
        return;
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint64_t x = i + (uint64_t)(4 * w + b);
            v |= (x < n ? (uint32_t)p[x] : 0u) << (8 * b);
        }
        out[w] = v;
    }
}

// one u32 of packed output (16 bases) per thread and turn; blockIdx.x = sequence, blockIdx.y = part of a long sequence
__global__ void pack_text_kernel(const unsigned char *__restrict__ text, const uint64_t *__restrict__ seq_off, const uint32_t *__restrict__ seq_len,
                                 const uint64_t *__restrict__ byte_off, uint8_t *__restrict__ packed) {
    const uint64_t s = blockIdx.x;
    const uint32_t len = seq_len[s];
    const unsigned char *src = text + seq_off[s];
    uint32_t *dst = (uint32_t *)(packed + byte_off[s]);  // (every sequence starts on a 4-byte boundary, pag_seqs)
    // the sequence's storage: ceil(len / 4) bytes rounded up to a multiple of 4, the bytes past the last base zero
    const uint32_t n_words = ((len + 3u) / 4u + 3u) / 4u;
    for (uint32_t w = blockIdx.y * blockDim.x + threadIdx.x; w < n_words; w += gridDim.y * blockDim.x) {
        uint32_t c[4];
        load16(src, (uint64_t)w * 16u, len, c);
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t i = w * 16u + (uint32_t)(4 * q + b);
                const uint32_t code = i < len ? base_code((c[q] >> (8 * b)) & 0xFFu) : 0u;
                v |= code << (2 * (4 * q + b));
            }
        dst[w] = v;
    }
}

// blockIdx.x = record; one u32 of classes (16 columns) per thread and turn
__global__ void classify_kernel(const unsigned char *__restrict__ text, const uint64_t *__restrict__ q_off, const uint32_t *__restrict__ q_len,
                                const uint64_t *__restrict__ r_off, const uint32_t *__restrict__ r_len, const uint64_t *__restrict__ diff_off,
                                uint32_t *__restrict__ diff, uint32_t *__restrict__ n_emit, uint32_t *__restrict__ n_radv) {
    const uint64_t rec = blockIdx.x;
    const uint32_t n = q_len[rec], rn = r_len[rec];
    const unsigned char *ql = text + q_off[rec], *rl = text + r_off[rec];
    uint32_t *out = diff + diff_off[rec];
    const uint32_t n_words = (n + 15u) / 16u;
    uint32_t emit = 0, radv = 0;
    for (uint32_t w = threadIdx.x; w < n_words; w += blockDim.x) {
        uint32_t qc[4], rc[4];
        load16(ql, (uint64_t)w * 16u, n, qc);
        load16(rl, (uint64_t)w * 16u, rn, rc);  // (past the end of a shorter reference row: NUL)
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t i = w * 16u + (uint32_t)(4 * q + b);
                if (i < n) {
                    const uint32_t a = (qc[q] >> (8 * b)) & 0xFFu, r = (rc[q] >> (8 * b)) & 0xFFu;
                    const uint32_t cls = a == '-' ? 1u : r == '-' ? 2u : a != r ? 3u : 0u;
                    v |= cls << (2 * (4 * q + b));
                    emit += cls != 1u;
                    radv += cls != 2u;
                }
            }
        out[w] = v;
    }
    emit = wave_sum(emit);
    radv = wave_sum(radv);
    if (lane_id() == 0) {
        if (emit) atomicAdd(&n_emit[rec], emit);
        if (radv) atomicAdd(&n_radv[rec], radv);
    }
}

// text in host memory -> a device buffer, through two pinned staging chunks filled by a few threads
int upload_text(const char *text, uint64_t bytes, unsigned char **dev, hipStream_t s) {
    *dev = nullptr;
    PAG_HIP_TRY(hipMalloc((void **)dev, bytes + 64));
    PAG_HIP_TRY(hipMemsetAsync(*dev + bytes, 0, 64, s));
    const size_t CH = 64u << 20;
    char *pin[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    auto drop = [&]() {
        for (int b = 0; b < 2; ++b) {
            if (pin[b]) hipHostFree(pin[b]);
            if (done[b]) hipEventDestroy(done[b]);
        }
    };
    for (int b = 0; b < 2; ++b) {
        if (hipHostMalloc((void **)&pin[b], CH, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&done[b], hipEventDisableTiming) != hipSuccess) {
            drop();  // (what the first turn of this loop took, and the device buffer: the caller sees no half-made upload)
            hipFree(*dev);
            *dev = nullptr;
            set_error("ingest: pinned staging buffers");
            return PAG_ENOMEM;
        }
    }
    const unsigned nthr = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    int rc = PAG_OK;
    for (uint64_t at = 0, c = 0; at < bytes && rc == PAG_OK; at += CH, ++c) {
        const int b = (int)(c & 1u);
        const size_t n = (size_t)std::min<uint64_t>(CH, bytes - at);
        if (c >= 2 && hipEventSynchronize(done[b]) != hipSuccess) rc = PAG_EFAULT;  // (the upload out of this buffer two chunks ago)
        std::vector<std::thread> pool;
        const size_t per = (n + nthr - 1) / nthr;
        for (unsigned t = 0; t < nthr; ++t) {
            const size_t lo = (size_t)t * per, hi = std::min(n, lo + per);
            if (lo < hi) pool.emplace_back([=]() { std::memcpy(pin[b] + lo, text + at + lo, hi - lo); });
        }
        for (auto &t : pool) t.join();
        if (hipMemcpyAsync(*dev + at, pin[b], n, hipMemcpyHostToDevice, s) != hipSuccess || hipEventRecord(done[b], s) != hipSuccess) rc = PAG_EFAULT;
    }
    if (hipStreamSynchronize(s) != hipSuccess) rc = PAG_EFAULT;
    drop();
    if (rc != PAG_OK) set_error("ingest: upload of the text failed");
    return rc;
}

struct DevTmp {  // small device arrays of one call
    std::vector<void *> p;
    ~DevTmp() {
        for (void *q : p) hipFree(q);
    }
    template <typename T>
    int put(const T *host, uint64_t n, T **out, hipStream_t s) {
        void *d = nullptr;
        PAG_HIP_TRY(hipMalloc(&d, std::max<uint64_t>(n, 1) * sizeof(T)));
        p.push_back(d);
        if (n) PAG_HIP_TRY(hipMemcpyAsync(d, host, n * sizeof(T), hipMemcpyHostToDevice, s));
        *out = (T *)d;
        return PAG_OK;
    }
};

}  // namespace
}  // namespace pagdev

using namespace pagdev;

extern "C" {

int pag_pack_text_seqs(const char *text, int text_on_device, uint64_t text_bytes, const uint64_t *seq_off, const uint32_t *seq_len, uint64_t n_seqs,
                       const uint64_t *byte_off, uint8_t *packed_dev, uint64_t packed_bytes, int device) {
    if ((!text && text_bytes) || (n_seqs && (!seq_off || !seq_len || !byte_off)) || !packed_dev) return PAG_EINVAL;
    if (!pag_device_available()) return PAG_ENODEV;  // (no CPU fallback)
    int rc = PAG_OK;
    PAG_HIP_TRY(hipSetDevice(device));
    uint32_t max_len = 0;
    for (uint64_t i = 0; i < n_seqs; ++i) {
        const uint64_t stored = (((uint64_t)seq_len[i] + 3) / 4 + 3) & ~3ull;
        if (seq_off[i] + seq_len[i] > text_bytes || (byte_off[i] & 3u) || byte_off[i] + stored > packed_bytes) {
            set_error("pag_pack_text_seqs: sequence %llu lies outside the text or the packed buffer", (unsigned long long)i);
            return PAG_EINVAL;
        }
        max_len = std::max(max_len, seq_len[i]);
    }
    if (n_seqs == 0) return PAG_OK;
    hipStream_t s = nullptr;
    unsigned char *dtext = (unsigned char *)text, *own = nullptr;
    if (!text_on_device) {
        if ((rc = upload_text(text, text_bytes, &own, s))) {
            if (own) hipFree(own);
            return rc;
        }
        dtext = own;
    }
    {
        DevTmp t;
        uint64_t *d_so, *d_bo;
        uint32_t *d_sl;
        if (!(rc = t.put(seq_off, n_seqs, &d_so, s)) && !(rc = t.put(seq_len, n_seqs, &d_sl, s)) && !(rc = t.put(byte_off, n_seqs, &d_bo, s))) {
            const uint32_t words = (max_len + 15u) / 16u;
            const unsigned parts = std::max(1u, std::min(1024u, (words + 4095u) / 4096u));
            // (2^31 - 1 blocks in x: more sequences than that would be more than the 2^32-coordinate space holds)
            pack_text_kernel<<<dim3((unsigned)n_seqs, parts), dim3(256), 0, s>>>(dtext, d_so, d_sl, d_bo, packed_dev);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
                set_error("pag_pack_text_seqs: kernel failed");
                rc = PAG_EFAULT;
            }
        }
    }
    if (own) hipFree(own);
    return rc;
}

int pag_classify_columns(const char *text, int text_on_device, uint64_t text_bytes, const uint64_t *q_off, const uint32_t *q_len, const uint64_t *r_off,
                         const uint32_t *r_len, const uint64_t *diff_off, uint64_t n_recs, uint32_t *diff_dev, uint64_t n_diff_words,
                         uint32_t *n_emit_dev, uint32_t *n_radv_dev, int device) {
    if ((!text && text_bytes) || (n_recs && (!q_off || !q_len || !r_off || !r_len || !diff_off)) || !diff_dev || !n_emit_dev || !n_radv_dev) return PAG_EINVAL;
    if (!pag_device_available()) return PAG_ENODEV;  // (no CPU fallback)
    int rc = PAG_OK;
    PAG_HIP_TRY(hipSetDevice(device));
    for (uint64_t i = 0; i < n_recs; ++i)
        if (q_off[i] + q_len[i] > text_bytes || r_off[i] + r_len[i] > text_bytes || diff_off[i] + ((uint64_t)q_len[i] + 15) / 16 > n_diff_words) {
            set_error("pag_classify_columns: record %llu lies outside the text or the class buffer", (unsigned long long)i);
            return PAG_EINVAL;
        }
    if (n_recs == 0) return PAG_OK;
    hipStream_t s = nullptr;
    unsigned char *dtext = (unsigned char *)text, *own = nullptr;
    if (!text_on_device) {
        if ((rc = upload_text(text, text_bytes, &own, s))) {
            if (own) hipFree(own);
            return rc;
        }
        dtext = own;
    }
    {
        DevTmp t;
        uint64_t *d_qo, *d_ro, *d_do;
        uint32_t *d_ql, *d_rl;
        if (!(rc = t.put(q_off, n_recs, &d_qo, s)) && !(rc = t.put(q_len, n_recs, &d_ql, s)) && !(rc = t.put(r_off, n_recs, &d_ro, s)) &&
            !(rc = t.put(r_len, n_recs, &d_rl, s)) && !(rc = t.put(diff_off, n_recs, &d_do, s))) {
            if (hipMemsetAsync(n_emit_dev, 0, n_recs * 4, s) != hipSuccess || hipMemsetAsync(n_radv_dev, 0, n_recs * 4, s) != hipSuccess) rc = PAG_EFAULT;
            classify_kernel<<<dim3((unsigned)n_recs), dim3(256), 0, s>>>(dtext, d_qo, d_ql, d_ro, d_rl, d_do, diff_dev, n_emit_dev, n_radv_dev);
            if (rc || hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
                set_error("pag_classify_columns: kernel failed");
                rc = PAG_EFAULT;
            }
        }
    }
    if (own) hipFree(own);
    return rc;
}

// ... with the results in HOST arrays (the class words and the two counts come back in one copy each): what a host parser
// that keeps its database in host memory calls (aln_db.cpp under PAGRAPH_DEVICE_INGEST=1)
int pag_classify_columns_host(const char *text, uint64_t text_bytes, const uint64_t *q_off, const uint32_t *q_len, const uint64_t *r_off,
                              const uint32_t *r_len, const uint64_t *diff_off, uint64_t n_recs, uint32_t *diff_host, uint64_t n_diff_words,
                              uint32_t *n_emit_host, uint32_t *n_radv_host, int device) {
    if (!diff_host || !n_emit_host || !n_radv_host) return PAG_EINVAL;
    if (!pag_device_available()) return PAG_ENODEV;
    PAG_HIP_TRY(hipSetDevice(device));
    uint32_t *d_diff = nullptr, *d_cnt = nullptr;
    PAG_HIP_TRY(hipMalloc((void **)&d_diff, (n_diff_words + 4) * 4));
    if (hipMalloc((void **)&d_cnt, (2 * n_recs + 2) * 4) != hipSuccess) {
        hipFree(d_diff);
        set_error("pag_classify_columns_host: out of device memory");
        return PAG_ENOMEM;
    }
    int rc = pag_classify_columns(text, 0, text_bytes, q_off, q_len, r_off, r_len, diff_off, n_recs, d_diff, n_diff_words, d_cnt, d_cnt + n_recs, device);
    if (rc == PAG_OK && (hipMemcpy(diff_host, d_diff, n_diff_words * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                         hipMemcpy(n_emit_host, d_cnt, n_recs * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                         hipMemcpy(n_radv_host, d_cnt + n_recs, n_recs * 4, hipMemcpyDeviceToHost) != hipSuccess)) {
        set_error("pag_classify_columns_host: copy back failed");
        rc = PAG_EFAULT;
    }
    hipFree(d_diff);
    hipFree(d_cnt);
    return rc;
}

}  // extern "C"
