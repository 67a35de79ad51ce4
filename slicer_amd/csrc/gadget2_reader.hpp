// gadget2_reader.hpp -- bulk reader of GADGET-2 format-2 sub-files (host side, no GPU code).
// Same on-disk contract as the reference reader (gadget2io.cpp:8-31 readHeader, :125-165
// fastforwardNVars/fastforwardToBlock, data.h:88-95 Block), but a block is read with one fread
// instead of three 4-byte stream reads per particle (gadget2io.cpp:200-202).
#pragma once
#include <cstdio>
#include <string>
#include <vector>

#include "slicer_types.hpp"

namespace slicer_amd {

class SnapshotFile {
public:
    ~SnapshotFile() { close(); }
    // readHeader semantics: try file_in, then file_in without its last two characters (".0").
    bool open(const std::string &file_in);
    void close();
    const Header &header() const { return hdr_; }
    const std::string &path() const { return path_; }
    // Reads the whole named block (searching forward from the block after HEAD); false if absent.
    bool read_block(const char *name4, std::vector<float> &out);
    // Per-type masses of the types with massarr == 0 (densitymaps.cpp:358-372): MASS in type order,
    // type 5 from BHMA.  mass[t] stays empty for types that use massarr.
    bool read_masses(std::vector<float> (&mass)[6]);
    // Streaming access: where a block's payload starts, and a positioned read of part of it.
    bool locate_block(const char *name4, long &offset, long &nbytes) { return find_block(name4, offset, nbytes); }
    bool read_at(long offset, void *dst, size_t bytes);

private:
    bool find_block(const char *name4, long &offset, long &nbytes);
    FILE *f_ = nullptr;
    Header hdr_{};
    std::string path_;
};

}  // namespace slicer_amd
