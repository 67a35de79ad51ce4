#include "gadget2_reader.hpp"

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <thread>
#include <vector>

#include <cstring>

namespace slicer_amd {

bool SnapshotFile::open(const std::string &file_in)
{
    close();
    path_ = file_in;
    f_ = fopen(path_.c_str(), "rb");
    if (!f_ && file_in.size() > 2) {  // gadget2io.cpp:14-17
        path_ = file_in.substr(0, file_in.size() - 2);
        f_ = fopen(path_.c_str(), "rb");
    }
    if (!f_)
        return false;
    int32_t blockheader[5];  // gadget2io.cpp:24-26
    if (fread(blockheader, sizeof blockheader, 1, f_) != 1 || fread(&hdr_, sizeof hdr_, 1, f_) != 1) {
        close();
        return false;
    }
    return true;
}

void SnapshotFile::close()
{
    if (f_)
        fclose(f_);
    f_ = nullptr;
}

bool SnapshotFile::find_block(const char *name4, long &offset, long &nbytes)
{
    if (!f_ || fseek(f_, 20 + 256, SEEK_SET) != 0)
        return false;
    Block b;
    while (fread(&b, sizeof b, 1, f_) == 1) {  // gadget2io.cpp:143-158
        if (memcmp(b.name, name4, 4) == 0) {
            offset = ftell(f_);
            nbytes = b.blocksize2;
            return true;
        }
        if (fseek(f_, b.blocksize2, SEEK_CUR) != 0)
            return false;
    }
    return false;
}

bool SnapshotFile::read_block(const char *name4, std::vector<float> &out)
{
    long off = 0, nbytes = 0;
    if (!find_block(name4, off, nbytes) || nbytes < 0)
        return false;
    out.resize((size_t)nbytes / sizeof(float));
    if (fseek(f_, off, SEEK_SET) != 0)
        return false;
    return out.empty() || fread(out.data(), sizeof(float), out.size(), f_) == out.size();
}

// Large reads are split over a few threads (pread on the same descriptor): one thread copies from the page cache at
// 3-4 GB/s, which is an order of magnitude below what the pinned staging buffer -> GPU path behind it takes.
// SLICER_AMD_READ_THREADS overrides the thread count (default: min(8, hardware threads)).
bool SnapshotFile::read_at(long offset, void *dst, size_t bytes)
{
    if (!f_)
        return false;
    if (bytes == 0)
        return true;
    static const unsigned nthreads = [] {
        const char *e = getenv("SLICER_AMD_READ_THREADS");
        unsigned n = e ? (unsigned)atoi(e) : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        return std::max(1u, std::min(n, 64u));
    }();
    constexpr size_t kMinPerThread = 4u << 20;
    const unsigned nt = (unsigned)std::min<size_t>(nthreads, bytes / kMinPerThread);
    if (nt <= 1) {
        if (fseek(f_, offset, SEEK_SET) != 0)
            return false;
        return fread(dst, 1, bytes, f_) == bytes;
    }
    const int fd = fileno(f_);
    std::atomic<bool> ok{true};
    auto work = [&](unsigned k) {
        const size_t per = (bytes / nt + 4095) & ~(size_t)4095;
        size_t lo = std::min(bytes, (size_t)k * per), hi = std::min(bytes, lo + per);
        if (k == nt - 1)
            hi = bytes;
        char *p = static_cast<char *>(dst) + lo;
        while (lo < hi) {
            const ssize_t r = pread(fd, p, hi - lo, (off_t)offset + (off_t)lo);
            if (r <= 0) {
                ok = false;
                return;
            }
            lo += (size_t)r;
            p += r;
        }
    };
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    for (unsigned k = 1; k < nt; k++)
        th.emplace_back(work, k);
    work(0);
    for (auto &t : th)
        t.join();
    return ok;
}

bool SnapshotFile::read_masses(std::vector<float> (&mass)[6])
{
    bool need = false;
    for (int t = 0; t < 6; t++) {
        mass[t].clear();
        need |= hdr_.npart[t] > 0 && hdr_.massarr[t] == 0;
    }
    if (!need)
        return true;
    std::vector<float> m;
    if (!read_block("MASS", m))
        return false;
    size_t off = 0;
    for (int t = 0; t < 6; t++) {
        if (!(hdr_.npart[t] > 0 && hdr_.massarr[t] == 0))
            continue;
        const size_t n = (size_t)hdr_.npart[t];
        if (t == 5) {  // densitymaps.cpp:361-365: skip npart[5] MASS entries, stream from BHMA
            std::vector<float> bh;
            if (!read_block("BHMA", bh) || bh.size() < n)
                return false;
            mass[5].assign(bh.begin(), bh.begin() + n);
        } else {
            if (off + n > m.size())
                return false;
            mass[t].assign(m.begin() + off, m.begin() + off + n);
        }
        off += n;
    }
    return true;
}

}  // namespace slicer_amd
