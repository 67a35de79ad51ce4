// fits_driver.cpp -- writes one plane with the C++ writeMaps twin; tests/test_fits.py compares the bytes with the
// Python writer (which is itself pinned byte-for-byte to libcfitsio).   usage: fits_driver <dir/> <partinplanes 0|1>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <valarray>

#include "fits_writer.hpp"

int main(int argc, char **argv)
{
    if (argc != 3)
        return 2;
    InputParams p{};
    p.npix = 16;
    p.fov = 2.0;
    p.simType = "Gadget";
    p.partinplanes = atoi(argv[2]) != 0;
    p.directory = argv[1];
    p.simulation = "sim";
    p.suffix = "test";
    p.snpix = "16";
    Header data{};
    data.h = 0.6774;
    data.om0 = 0.3089;
    data.oml = 0.6911;
    const double m[6] = {0.0, 0.0123456789, 0, 1e-30, 2.5e10, 0.5};
    for (int i = 0; i < 6; i++)
        data.massarr[i] = m[i];
    Lens lens{};
    lens.nplanes = 1;
    lens.ld = {123.456789};
    lens.ld2 = {223.4};
    std::valarray<float> tot(256), toti[6];
    for (int i = 0; i < 256; i++)
        tot[i] = (float)i * 0.37f;
    for (int t = 0; t < 6; t++) {
        toti[t].resize(256);
        for (int i = 0; i < 256; i++)
            toti[t][i] = (float)(t * 1000 + i);
    }
    int ntot[6] = {0, 123456, 0, 7, 0, 2147483647};
    try {
        writeMaps(p, data, lens, 0, 0.512345678901234567, "007", "16", tot, toti, ntot, 0);
    } catch (const std::runtime_error &) {
        return 1;
    }
    return 0;
}
