// fits_writer.hpp -- writeMaps / fileOutput of the reference (densitymaps.cpp:530-649) without CCfits:
// a plain 2880-byte-block writer whose output is byte-identical to what cfitsio 3.47 produces for the
// reference's CCfits call sequence (pinned in tests/test_fits.py against libcfitsio; the Python twin is
// slicer_amd/fits.py and tests/test_cpp_adapter.py compares the two byte for byte).
#pragma once
#include <string>
#include <valarray>

#include "slicer_types.hpp"

// Same signatures as the reference.  writeMaps throws std::runtime_error where CCfits would throw
// FITS::CantCreate (file exists / cannot be created); slicer-v2.cpp:220-229 turns that into MPI_Abort.
void writeMaps(InputParams &p, Header &data, Lens &lens, int isnap, double zsim, std::string snappl, std::string snpix,
               std::valarray<float> &mapxytotrecv, std::valarray<float> (&mapxytotirecv)[6], int (&ntotxyi)[6],
               int myid);
std::string fileOutput(InputParams p, std::string snappl, int label = 0);

namespace slicer_amd {
struct FitsKey {
    std::string name;
    bool is_int;
    long ival;
    double dval;
    std::string comment;
};
std::string fits_card(const FitsKey &k);
// returns false if the file exists or cannot be written
bool fits_write_image(const std::string &path, const float *image, int npix, const FitsKey *keys, int nkeys);
}  // namespace slicer_amd
