// blob.h -- host-side reader of the native "RVCW" weight blob (format: obs_rvc_amd/weights.py).
// Replaces the ONNX session factory of the reference (rvc/src/models.rs:7-76) on the load path.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace rvc {

struct BlobTensor {
    const float *data = nullptr;
    std::vector<int> dims;
    size_t nelem = 0;
};

class Blob {
public:
    explicit Blob(const std::string &path)
    {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + path);
        fseek(f, 0, SEEK_END);
        long len = ftell(f);
        fseek(f, 0, SEEK_SET);
        raw_.resize((size_t)len);
        size_t got = fread(raw_.data(), 1, (size_t)len, f);
        fclose(f);
        if (got != (size_t)len) throw std::runtime_error("short read " + path);
        if (len < 24 || memcmp(raw_.data(), "RVCW0001", 8) != 0) throw std::runtime_error("not an RVCW blob: " + path);
        uint32_t n_cfg, n_ten;
        uint64_t data_off;
        memcpy(&n_cfg, raw_.data() + 8, 4);
        memcpy(&n_ten, raw_.data() + 12, 4);
        memcpy(&data_off, raw_.data() + 16, 8);
        size_t p = 24;
        for (uint32_t i = 0; i < n_cfg; i++) {
            char name[49] = {0};
            double v;
            memcpy(name, raw_.data() + p, 48);
            memcpy(&v, raw_.data() + p + 48, 8);
            cfg_[name] = v;
            p += 56;
        }
        for (uint32_t i = 0; i < n_ten; i++) {
            char name[97] = {0};
            uint32_t ndim, dims[5];
            uint64_t off, nelem;
            memcpy(name, raw_.data() + p, 96);
            memcpy(&ndim, raw_.data() + p + 96, 4);
            memcpy(dims, raw_.data() + p + 100, 20);
            memcpy(&off, raw_.data() + p + 120, 8);
            memcpy(&nelem, raw_.data() + p + 128, 8);
            p += 136;
            if (data_off + off + nelem * 4 > raw_.size()) throw std::runtime_error("corrupt blob (tensor out of range): " + path);
            BlobTensor t;
            t.data = reinterpret_cast<const float *>(raw_.data() + data_off + off);
            t.dims.assign(dims, dims + ndim);
            t.nelem = nelem;
            ten_[name] = t;
        }
        path_ = path;
    }
    int icfg(const std::string &k) const
    {
        auto it = cfg_.find(k);
        if (it == cfg_.end()) throw std::runtime_error("blob " + path_ + ": missing cfg " + k);
        return (int)it->second;
    }
    bool has(const std::string &k) const { return ten_.count(k) != 0; }
    const BlobTensor &t(const std::string &k) const
    {
        auto it = ten_.find(k);
        if (it == ten_.end()) throw std::runtime_error("blob " + path_ + ": missing tensor " + k);
        return it->second;
    }
    const float *w(const std::string &k) const { return t(k).data; }
    size_t bytes() const { return raw_.size(); }

private:
    std::vector<unsigned char> raw_;
    std::map<std::string, double> cfg_;
    std::map<std::string, BlobTensor> ten_;
    std::string path_;
};

static inline std::string fmt(const char *f, int a) { char b[160]; snprintf(b, sizeof b, f, a); return b; }
static inline std::string fmt(const char *f, int a, int c) { char b[160]; snprintf(b, sizeof b, f, a, c); return b; }
static inline std::string fmt(const char *f, int a, int c, int d) { char b[160]; snprintf(b, sizeof b, f, a, c, d); return b; }

}  // namespace rvc
