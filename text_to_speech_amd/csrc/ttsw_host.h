// ttsw_host.h -- the host-only part of the C ABI's weight loading: the TTSW container parser and the tensor-shape checks.
// Pure C++17 (no HIP header, no GPU call), so that this code -- which reads UNTRUSTED files -- is also built for the CPU with
// -fsanitize=address,undefined (csrc/host_check.cpp, tests/test_host_sanitizer.py).  The reference's loader trusts its
// checkpoint files (/root/reference/custom_train_objects/checkpoint_manager.py:169-215); a C loader cannot.
#pragma once
#include <stdint.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/tts_hip.h"

#ifndef TTS_HOST_TENSOR_DEFINED
#define TTS_HOST_TENSOR_DEFINED
struct HostTensor {
    std::vector<int64_t> dims;
    std::vector<float> data;
    size_t numel() const {
        size_t n = 1;
        for (auto d : dims) n *= (size_t)d;
        return n;
    }
};
#endif

// numel of `dims` if every dim is positive and the product stays below kMaxTensorElems (2^34 floats = 64 GiB); 0 otherwise
static inline size_t checked_numel(const std::vector<int64_t>& dims) {
    constexpr uint64_t kMaxTensorElems = 1ull << 34;
    uint64_t n = 1;
    for (int64_t d : dims) {
        if (d <= 0 || (uint64_t)d > kMaxTensorElems) return 0;
        n *= (uint64_t)d;
        if (n > kMaxTensorElems) return 0;
    }
    return (size_t)n;
}

// Parses a TTSW file into `out` (may be null: validation only).  Nothing in the file is trusted: the entry count, name
// lengths, dims and payload ranges are all checked against the file size before anything is allocated from them.
static inline int parse_ttsw(const char* path, std::map<std::string, HostTensor>* out, std::string* err) {
    FILE* f = fopen(path, "rb");
    if (!f) {
        *err = std::string("cannot open ") + path;
        return TTS_HIP_EIO;
    }
    auto fail = [&](const char* what) {
        fclose(f);
        *err = std::string(path) + ": " + what;
        return TTS_HIP_EIO;
    };
    if (fseek(f, 0, SEEK_END) != 0) return fail("cannot seek");
    const long long fsize = ftell(f);
    if (fsize < 12 || fseek(f, 0, SEEK_SET) != 0) return fail("not a TTSW file");
    char magic[4];
    uint32_t ver = 0, n = 0;
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "TTSW", 4) != 0) return fail("not a TTSW file");
    if (fread(&ver, 4, 1, f) != 1 || fread(&n, 4, 1, f) != 1 || ver != 1) return fail("unsupported version");
    // an entry header is at least 4 + 1 + 4 + 8 + 8 + 8 = 33 bytes
    if ((unsigned long long)n * 33ull > (unsigned long long)fsize) return fail("entry count exceeds file size");
    struct Ent {
        std::string name;
        std::vector<int64_t> dims;
        uint64_t off, nbytes;
    };
    try {
        std::vector<Ent> ents(n);
        for (auto& en : ents) {
            uint32_t ln = 0, nd = 0;
            if (fread(&ln, 4, 1, f) != 1 || ln == 0 || ln > 4096) return fail("bad name length");
            en.name.resize(ln);
            if (fread(&en.name[0], 1, ln, f) != ln) return fail("truncated header");
            if (fread(&nd, 4, 1, f) != 1 || nd == 0 || nd > 8) return fail("bad ndim");
            en.dims.resize(nd);
            if (fread(en.dims.data(), 8, nd, f) != nd) return fail("truncated header");
            if (fread(&en.off, 8, 1, f) != 1 || fread(&en.nbytes, 8, 1, f) != 1) return fail("truncated header");
            const size_t numel = checked_numel(en.dims);
            if (!numel) return fail("non-positive or oversized dim");
            if ((uint64_t)numel * sizeof(float) != en.nbytes) return fail("size mismatch");
            if (en.off > (uint64_t)fsize || en.nbytes > (uint64_t)fsize - en.off) return fail("payload outside the file");
        }
        for (auto& en : ents) {
            if (!out) continue;
            HostTensor t;
            t.dims = en.dims;
            t.data.resize(en.nbytes / sizeof(float));
            if (fseek(f, (long)en.off, SEEK_SET) != 0 || fread(t.data.data(), 1, en.nbytes, f) != en.nbytes)
                return fail("truncated payload");
            (*out)[en.name] = std::move(t);
        }
    } catch (const std::exception& ex) {
        fclose(f);
        *err = std::string(path) + ": " + ex.what();
        return TTS_HIP_ENOMEM;
    }
    fclose(f);
    return TTS_HIP_OK;
}

