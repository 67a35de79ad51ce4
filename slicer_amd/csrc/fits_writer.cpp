#include "fits_writer.hpp"

#include <sys/stat.h>

#include <cstdio>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <vector>

namespace slicer_amd {

static std::string fmt_double(double v)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.15G", v);
    std::string s(buf);
    if (s.find('.') == std::string::npos && s.find('N') == std::string::npos && s.find('I') == std::string::npos) {
        size_t e = s.find('E');
        if (e != std::string::npos)
            s.insert(e, ".0");  // cfitsio: "1E-30" -> "1.0E-30"
        else
            s += ".";           // "2" -> "2."
    }
    return s;
}

std::string fits_card(const FitsKey &k)
{
    std::string name = k.name;
    for (auto &c : name)
        c = (char)toupper((unsigned char)c);
    const std::string val = k.is_int ? std::to_string(k.ival) : fmt_double(k.dval);
    std::string head;
    if (name.size() <= 8) {
        head = name;
        head.resize(8, ' ');
        head += "= ";
    } else {
        head = "HIERARCH " + name + " = ";  // ESO convention, as cfitsio writes long keyword names
    }
    std::string card = head;
    if (head.size() + val.size() < 30)
        card.append(30 - head.size() - val.size(), ' ');  // value right-justified to column 30
    card += val + " / " + k.comment;
    card.resize(80, ' ');
    return card;
}

bool fits_write_image(const std::string &path, const float *image, int npix, const FitsKey *keys, int nkeys)
{
    struct stat st;
    if (stat(path.c_str(), &st) == 0)
        return false;  // CCfits refuses to overwrite (no leading '!')
    std::string hdr;
    auto add = [&](std::string c) {
        c.resize(80, ' ');
        hdr += c;
    };
    char buf[96];
    add("SIMPLE  =                    T / file does conform to FITS standard");
    add("BITPIX  =                  -32 / number of bits per data pixel");
    add("NAXIS   =                    2 / number of data axes");
    snprintf(buf, sizeof buf, "NAXIS1  = %20d / length of data axis 1", npix);
    add(buf);
    snprintf(buf, sizeof buf, "NAXIS2  = %20d / length of data axis 2", npix);
    add(buf);
    add("EXTEND  =                    T / FITS dataset may contain extensions");
    add("COMMENT   FITS (Flexible Image Transport System) format is defined in 'Astronomy");
    add("COMMENT   and Astrophysics', volume 376, page 359; bibcode: 2001A&A...376..359H");
    for (int i = 0; i < nkeys; i++)
        hdr += fits_card(keys[i]);
    add("END");
    hdr.append((2880 - hdr.size() % 2880) % 2880, ' ');

    const size_t n = (size_t)npix * (size_t)npix;
    std::vector<unsigned char> data(n * 4 + (2880 - (n * 4) % 2880) % 2880, 0);
    for (size_t i = 0; i < n; i++) {  // big-endian IEEE f32
        uint32_t u;
        memcpy(&u, &image[i], 4);
        data[4 * i + 0] = (unsigned char)(u >> 24);
        data[4 * i + 1] = (unsigned char)(u >> 16);
        data[4 * i + 2] = (unsigned char)(u >> 8);
        data[4 * i + 3] = (unsigned char)u;
    }
    FILE *f = fopen(path.c_str(), "wb");
    if (!f)
        return false;
    bool ok = fwrite(hdr.data(), 1, hdr.size(), f) == hdr.size() && fwrite(data.data(), 1, data.size(), f) == data.size();
    ok = (fclose(f) == 0) && ok;
    return ok;
}

}  // namespace slicer_amd

using slicer_amd::FitsKey;

std::string fileOutput(InputParams p, std::string snappl, int label)
{
    if (p.simType == "Gadget" && p.partinplanes == false)
        return p.directory + p.simulation + "." + snappl + ".plane_" + p.snpix + "_" + p.suffix + ".fits";
    else if (p.simType == "Gadget" && p.partinplanes == true)
        return p.directory + p.simulation + "." + snappl + ".ptype" + std::to_string(label) + "_plane_" + p.snpix + "_" +
               p.suffix + ".fits";
    throw std::invalid_argument("Output name format not recognized");
}

static std::vector<FitsKey> plane_keys(InputParams &p, Header &data, Lens &lens, int isnap, double zsim,
                                       const int (&ntotxyi)[6], int ptype)
{
    std::vector<FitsKey> k;
    auto d = [&](const std::string &n, double v, const std::string &c = " ") { k.push_back({n, false, 0, v, c}); };
    auto i = [&](const std::string &n, long v) { k.push_back({n, true, v, 0.0, " "}); };
    d("REDSHIFT", zsim);
    d("PHYSICALSIZE", p.fov);
    d("PIXELUNIT", 1.e+10 / data.h, "Mass unit in M_Sun");
    d("DlLOW", lens.ld[isnap] / data.h, "comoving distance in Mpc");
    d("DlUP", lens.ld2[isnap] / data.h, "comoving distance in Mpc");
    if (ptype < 0)
        for (int t = 0; t < 6; t++)
            i("nparttype" + std::to_string(t), ntotxyi[t]);
    else
        i("nparttype0", ntotxyi[ptype]);
    d("HUBBLE", data.h);
    d("OMEGAMATTER", data.om0);
    d("OMEGALAMBDA", data.oml);
    if (ptype < 0)
        for (int t = 0; t < 6; t++)
            d("m" + std::to_string(t), data.massarr[t]);
    else
        d("m" + std::to_string(ptype), data.massarr[ptype]);
    return k;
}

void writeMaps(InputParams &p, Header &data, Lens &lens, int isnap, double zsim, std::string snappl, std::string,
               std::valarray<float> &mapxytotrecv, std::valarray<float> (&mapxytotirecv)[6], int (&ntotxyi)[6],
               int myid)
{
    if (myid != 0)
        return;
    if (p.partinplanes == false) {
        const std::string fileoutput = fileOutput(p, snappl);
        std::cout << "Saving the maps on: " << fileoutput << std::endl;
        auto keys = plane_keys(p, data, lens, isnap, zsim, ntotxyi, -1);
        if (!slicer_amd::fits_write_image(fileoutput, &mapxytotrecv[0], p.npix, keys.data(), (int)keys.size())) {
            std::cerr << "It was not possible to create the map: " << fileoutput << std::endl;
            throw std::runtime_error("FITS::CantCreate");
        }
    } else {
        for (int t = 0; t < 6; t++) {
            if (ntotxyi[t] > 0) {
                const std::string fileoutput = fileOutput(p, snappl, t);
                auto keys = plane_keys(p, data, lens, isnap, zsim, ntotxyi, t);
                if (!slicer_amd::fits_write_image(fileoutput, &mapxytotirecv[t][0], p.npix, keys.data(),
                                                  (int)keys.size())) {
                    std::cerr << "It was not possible to create the map: " << fileoutput << std::endl;
                    throw std::runtime_error("FITS::CantCreate");
                }
            }
        }
    }
}
