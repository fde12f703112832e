// Host side of the text reader (C ABI hml_text_*, include/hml.h): stages the stream in pinned chunks cut at
// whitespace, runs the kernels of hml_k_text.h on them, and resolves the tokens the device reported as
// irregular with the real stream extraction of libstdc++ - the statement the reference executes for every
// value (reference src/wavelet.hpp:131) - so the result is what `while ( input >> v )` yields, including where
// it stops.  Two slots: while the caller fills one pinned buffer, the GPU works on the other.
// Included by hml_capi.hip (uses its set_err / HIPCHK).
#ifndef HML_TEXT_READER_HPP
#define HML_TEXT_READER_HPP

#include <istream>
#include <streambuf>

#include "hml_k_text.h"

// float vector whose resize() leaves new elements uninitialised: every element is overwritten by the device-to-host
// copy, and touching 400 MB twice is a visible share of reading 10^8 values
template <class T>
struct hml_noinit_alloc : std::allocator<T> {
    template <class U> struct rebind { typedef hml_noinit_alloc<U> other; };
    template <class U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
typedef std::vector<float, hml_noinit_alloc<float>> hml_fvec;

#define HML_TEXT_DEFAULT_CHUNK (64u << 20)
#define HML_TEXT_IRR_CAP 65536u

struct hml_text {
    int device = 0;
    hipStream_t stream = nullptr;
    uint64_t chunk = 0;                 // staging capacity in bytes
    char* h_in[2] = {nullptr, nullptr}; // pinned staging buffers
    hml_text_meta* h_meta = nullptr;    // pinned, 2 entries
    uint8_t* d_text[2] = {nullptr, nullptr};
    float* d_vals[2] = {nullptr, nullptr};
    hml_text_meta* d_meta = nullptr;    // 2 entries
    hml_text_irr* d_irr[2] = {nullptr, nullptr};
    uint32_t *d_tile_count = nullptr, *d_tile_base = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr};
    bool in_flight[2] = {false, false};
    uint64_t sub_len[2] = {0, 0};       // bytes of the submitted chunk in each slot
    int cur = 0;                        // slot being filled
    uint64_t fill = 0;                  // bytes in h_in[cur]
    bool finished = false, stopped = false;
    hml_fvec values;
    uint64_t bytes_in = 0, irregular_tokens = 0, host_chunks = 0;
};

namespace {

struct hml_membuf : std::streambuf {
    hml_membuf(const char* b, const char* e) { setg(const_cast<char*>(b), const_cast<char*>(b), const_cast<char*>(e)); }
};

// `while ( input >> v )` over [b, e): appends the values; false if an extraction failed (the reference's reader
// stops there), true if the text was consumed to its end
template <class Vec>
bool hml_text_host_extract(const char* b, const char* e, Vec& out) {
    hml_membuf mb(b, e);
    std::istream is(&mb);
    float v = 0;
    for (;;) {
        int c;
        while ((c = mb.sgetc()) != std::streambuf::traits_type::eof() && hml_is_space((uint32_t)(unsigned char)c)) mb.sbumpc();
        if (c == std::streambuf::traits_type::eof()) return true;
        if (!(is >> v)) return false;
        out.push_back(v);
    }
}

void hml_text_free(hml_text* p) {
    for (int j = 0; j < 2; ++j) {
        if (p->h_in[j]) hipHostFree(p->h_in[j]);
        if (p->d_text[j]) hipFree(p->d_text[j]);
        if (p->d_vals[j]) hipFree(p->d_vals[j]);
        if (p->d_irr[j]) hipFree(p->d_irr[j]);
        if (p->done[j]) hipEventDestroy(p->done[j]);
    }
    if (p->h_meta) hipHostFree(p->h_meta);
    if (p->d_meta) hipFree(p->d_meta);
    if (p->d_tile_count) hipFree(p->d_tile_count);
    if (p->d_tile_base) hipFree(p->d_tile_base);
    if (p->stream) hipStreamDestroy(p->stream);
}

// collects the values of the chunk in `slot` (submitted earlier) and resolves its irregular tokens
int hml_text_retire(hml_text* p, int slot) {
    if (!p->in_flight[slot]) return 0;
    p->in_flight[slot] = false;
    HIPCHK(hipEventSynchronize(p->done[slot]));
    if (p->stopped) return 0;
    const hml_text_meta m = p->h_meta[slot];
    const char* text = p->h_in[slot];
    const uint64_t len = p->sub_len[slot];
    if (m.irregular > HML_TEXT_IRR_CAP) {
        // more irregular tokens than the list holds: the whole chunk goes through the stream extraction
        p->host_chunks++;
        if (!hml_text_host_extract(text, text + len, p->values)) p->stopped = true;
        return 0;
    }
    const size_t at = p->values.size();
    p->values.resize(at + m.tokens);
    if (m.tokens) HIPCHK(hipMemcpyAsync(p->values.data() + at, p->d_vals[slot], (size_t)m.tokens * sizeof(float), hipMemcpyDeviceToHost, p->stream));
    std::vector<hml_text_irr> irr(m.irregular);
    if (m.irregular) HIPCHK(hipMemcpyAsync(irr.data(), p->d_irr[slot], irr.size() * sizeof(hml_text_irr), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    if (irr.empty()) return 0;
    p->irregular_tokens += irr.size();
    std::sort(irr.begin(), irr.end(), [](const hml_text_irr& a, const hml_text_irr& b) { return a.token < b.token; });
    // pass 1: every irregular token that yields exactly one value is patched in place
    bool simple = true;
    std::vector<std::vector<float>> got(irr.size());
    std::vector<char> clean(irr.size());
    for (size_t i = 0; i < irr.size(); ++i) {
        const char* b = text + irr[i].offset;
        const char* e = b;
        while (e < text + len && !hml_is_space((uint32_t)(unsigned char)*e)) ++e;
        clean[i] = hml_text_host_extract(b, e, got[i]);
        if (!clean[i] || got[i].size() != 1) simple = false;
        if (!clean[i]) { got.resize(i + 1); clean.resize(i + 1); irr.resize(i + 1); break; }   // nothing after a failure counts
    }
    if (simple) {
        for (size_t i = 0; i < irr.size(); ++i) p->values[at + irr[i].token] = got[i][0];
        return 0;
    }
    // general case: a token gave several values ("1.5-3"), none, or the extraction failed inside it
    const hml_fvec chunk(p->values.begin() + at, p->values.end());
    p->values.resize(at);
    size_t from = 0;
    for (size_t i = 0; i < irr.size(); ++i) {
        p->values.insert(p->values.end(), chunk.begin() + from, chunk.begin() + irr[i].token);
        p->values.insert(p->values.end(), got[i].begin(), got[i].end());
        from = (size_t)irr[i].token + 1;
        if (!clean[i]) { p->stopped = true; return 0; }
    }
    p->values.insert(p->values.end(), chunk.begin() + from, chunk.end());
    return 0;
}

// sends the first `len` bytes of slot `slot` to the GPU and enqueues the three kernels
int hml_text_submit(hml_text* p, int slot, uint64_t len) {
    p->sub_len[slot] = len;
    if (len == 0 || p->stopped) return 0;
    const uint32_t n_tiles = (uint32_t)((len + HML_TEXT_TILE - 1) / HML_TEXT_TILE);
    const uint64_t padded = ((uint64_t)n_tiles + 1) * HML_TEXT_TILE;
    HIPCHK(hipMemcpyAsync(p->d_text[slot], p->h_in[slot], len, hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipMemsetAsync(p->d_text[slot] + len, ' ', padded - len, p->stream));
    HIPCHK(hipMemsetAsync(p->d_meta + slot, 0, sizeof(hml_text_meta), p->stream));
    hipLaunchKernelGGL(hml_k_text_count, dim3(n_tiles), dim3(256), 0, p->stream, p->d_text[slot], n_tiles, p->d_tile_count);
    hipLaunchKernelGGL(hml_k_text_scan, dim3(1), dim3(1024), 0, p->stream, p->d_tile_count, n_tiles, p->d_tile_base, p->d_meta + slot);
    hipLaunchKernelGGL(hml_k_text_parse, dim3(n_tiles), dim3(256), 0, p->stream, p->d_text[slot], n_tiles, p->d_tile_base,
                       p->d_vals[slot], p->d_meta + slot, p->d_irr[slot], HML_TEXT_IRR_CAP);
    KLAUNCH_CHECK();
    HIPCHK(hipMemcpyAsync(p->h_meta + slot, p->d_meta + slot, sizeof(hml_text_meta), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipEventRecord(p->done[slot], p->stream));
    p->in_flight[slot] = true;
    return 0;
}

// the staging buffer is full (or the stream ended): cut at the last whitespace, submit, move the rest over
int hml_text_flush(hml_text* p, bool final) {
    const int slot = p->cur, other = 1 - slot;
    char* buf = p->h_in[slot];
    uint64_t cut = p->fill;
    if (!final) {
        while (cut > 0 && !hml_is_space((uint32_t)(unsigned char)buf[cut - 1])) --cut;
        if (cut == 0) return set_err(HML_ERR_ARG, "a single input token exceeds the text staging buffer");
    }
    if (int r = hml_text_submit(p, slot, cut)) return r;
    if (int r = hml_text_retire(p, other)) return r;          // the chunk before: its buffer is needed now
    const uint64_t rest = p->fill - cut;
    if (rest) memcpy(p->h_in[other], buf + cut, rest);
    p->cur = other;
    p->fill = rest;
    return 0;
}

}  // namespace

extern "C" {

int hml_text_open(hml_text** out, int device, uint64_t chunk_bytes) {
    if (!out) return set_err(HML_ERR_ARG, "null output pointer");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (n <= 0) return set_err(HML_ERR_HIP, "no HIP device available: the MI355X kernels cannot run (there is no CPU fallback)");
    if (device < 0 || device >= n) return set_err(HML_ERR_ARG, "device index out of range");
    if (chunk_bytes == 0) chunk_bytes = HML_TEXT_DEFAULT_CHUNK;
    if (chunk_bytes < 256 || chunk_bytes > (256u << 20)) return set_err(HML_ERR_ARG, "text chunk size must be in [256 B, 256 MiB]");
    HIPCHK(hipSetDevice(device));
    hml_text* p = new hml_text();
    p->device = device;
    p->chunk = chunk_bytes;
    const uint64_t tiles = (chunk_bytes + HML_TEXT_TILE - 1) / HML_TEXT_TILE;
    int rc = [&]() -> int {
        HIPCHK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        HIPCHK(hipHostMalloc(&p->h_meta, 2 * sizeof(hml_text_meta), hipHostMallocDefault));
        HIPCHK(hipMalloc(&p->d_meta, 2 * sizeof(hml_text_meta)));
        HIPCHK(hipMalloc(&p->d_tile_count, tiles * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&p->d_tile_base, tiles * sizeof(uint32_t)));
        for (int j = 0; j < 2; ++j) {
            HIPCHK(hipHostMalloc(&p->h_in[j], chunk_bytes, hipHostMallocDefault));
            HIPCHK(hipMalloc(&p->d_text[j], (tiles + 1) * HML_TEXT_TILE));
            HIPCHK(hipMalloc(&p->d_vals[j], (chunk_bytes / 2 + 1) * sizeof(float)));
            HIPCHK(hipMalloc(&p->d_irr[j], HML_TEXT_IRR_CAP * sizeof(hml_text_irr)));
            HIPCHK(hipEventCreateWithFlags(&p->done[j], hipEventDisableTiming));
        }
        return 0;
    }();
    if (rc) { hml_text_free(p); delete p; return rc; }
    *out = p;
    return 0;
}

void hml_text_close(hml_text* p) {
    if (!p) return;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    hml_text_free(p);
    delete p;
}

int hml_text_buffer(hml_text* p, char** buf, uint64_t* capacity) {
    if (!p || !buf || !capacity) return set_err(HML_ERR_ARG, "null argument");
    if (p->finished) return set_err(HML_ERR_ARG, "text reader already finished");
    HIPCHK(hipSetDevice(p->device));
    if (p->fill == p->chunk)
        if (int r = hml_text_flush(p, false)) return r;
    *buf = p->h_in[p->cur] + p->fill;
    *capacity = p->chunk - p->fill;
    return 0;
}

int hml_text_commit(hml_text* p, uint64_t nbytes) {
    if (!p) return set_err(HML_ERR_ARG, "null argument");
    if (p->finished) return set_err(HML_ERR_ARG, "text reader already finished");
    if (nbytes > p->chunk - p->fill) return set_err(HML_ERR_ARG, "commit exceeds the staging buffer");
    p->fill += nbytes;
    p->bytes_in += nbytes;
    return 0;
}

int hml_text_feed(hml_text* p, const char* bytes, uint64_t n) {
    if (!p || (!bytes && n)) return set_err(HML_ERR_ARG, "null argument");
    while (n) {
        char* buf; uint64_t cap;
        if (int r = hml_text_buffer(p, &buf, &cap)) return r;
        const uint64_t k = std::min(cap, n);
        memcpy(buf, bytes, k);
        if (int r = hml_text_commit(p, k)) return r;
        bytes += k; n -= k;
    }
    return 0;
}

int hml_text_finish(hml_text* p, uint64_t* n_values, int* stopped) {
    if (!p) return set_err(HML_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(p->device));
    if (!p->finished) {
        if (int r = hml_text_flush(p, true)) return r;     // submits the rest, retires the chunk before it
        if (int r = hml_text_retire(p, 1 - p->cur)) return r;   // flush switched slots: this is the last chunk
        p->finished = true;
    }
    if (n_values) *n_values = p->values.size();
    if (stopped) *stopped = p->stopped ? 1 : 0;
    return 0;
}

int hml_text_values(hml_text* p, float* out) {
    if (!p || !p->finished) return set_err(HML_ERR_ARG, "text reader not finished");
    if (!p->values.empty()) memcpy(out, p->values.data(), p->values.size() * sizeof(float));
    return 0;
}

int hml_text_reserve(hml_text* p, uint64_t n_values) {
    if (!p) return set_err(HML_ERR_ARG, "null argument");
    try { p->values.reserve(n_values); } catch (...) { return set_err(HML_ERR_ARG, "cannot reserve that many values"); }
    return 0;
}

int hml_text_counters(hml_text* p, uint64_t* bytes_in, uint64_t* irregular_tokens, uint64_t* host_chunks) {
    if (!p) return set_err(HML_ERR_ARG, "null argument");
    if (bytes_in) *bytes_in = p->bytes_in;
    if (irregular_tokens) *irregular_tokens = p->irregular_tokens;
    if (host_chunks) *host_chunks = p->host_chunks;
    return 0;
}

}  // extern "C"

#endif
