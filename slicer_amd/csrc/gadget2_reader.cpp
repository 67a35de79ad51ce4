#include "gadget2_reader.hpp"

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

bool SnapshotFile::read_at(long offset, void *dst, size_t bytes)
{
    if (!f_ || fseek(f_, offset, SEEK_SET) != 0)
        return false;
    return bytes == 0 || fread(dst, 1, bytes, f_) == bytes;
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
