// slicer_types.hpp -- the reference's path-level structs, for builds outside the SLICER tree.
//
// When the adapter is compiled inside SLICER (-DSLICER_AMD_REFERENCE_HEADERS) the
// reference's own data.h supplies these types and this header supplies nothing.  Stand-alone, the
// definitions below repeat the member names, order and types of data.h:29-131 so that code written
// against the reference (slicer-v2.cpp:138-229) compiles unchanged against this repository.
#pragma once
#ifdef SLICER_AMD_REFERENCE_HEADERS
#include "data.h"
#else
#include <cstdint>
#include <string>
#include <vector>

struct gsl_spline;        // opaque here: createDensityMaps never dereferences its four GSL arguments
struct gsl_interp_accel;  // (densitymaps.cpp:419-524 does not use them)

struct InputParams {  // data.h:29-49
    int npix;
    double zs;
    double Ds;
    double fov;
    bool hydro;
    std::string simType;
    double rgrid;
    std::string filredshiftlist;
    std::string pathsnap;
    std::string simulation;
    int seedcenter, seedface, seedsign;
    bool partinplanes;
    std::string directory;
    std::string suffix;
    int snopt;
    std::string snpix;
    bool physical;
    double w;
};

struct Header {  // data.h:59-79 -- the 256-byte GADGET-2 header
    int32_t npart[6];
    double massarr[6];
    double time;
    double redshift;
    int32_t flag_sfr;
    int32_t flag_feedback;
    uint32_t npartTotal[6];
    int32_t flag_cooling;
    int32_t numfiles;
    double boxsize;
    double om0;
    double oml;
    double h;
    int32_t flag_sage;
    int32_t flag_metals;
    int32_t nTotalHW[6];
    int32_t flag_entropy;
    int32_t la[14];
};
static_assert(sizeof(Header) == 256, "GADGET-2 header must be 256 bytes");

struct Block {  // data.h:88-95 -- format-2 block framing as the reference reads it
    int32_t blocksize1;
    int8_t alignment[4];
    char name[4];
    int8_t padding[8];
    int32_t blocksize2;
};
static_assert(sizeof(Block) == 24, "format-2 block record must be 24 bytes");

struct Lens {  // data.h:104-117
    int nplanes;
    std::vector<int> replication;
    std::vector<int> pll;
    std::vector<std::string> fromsnap;
    std::vector<int> fromsnapi;
    std::vector<double> zsimlens;
    std::vector<double> ld;
    std::vector<double> ld2;
    std::vector<double> zfromsnap;
    std::vector<bool> randomize;
    std::vector<int> nrepperp;
};

struct Random {  // data.h:126-131
    std::vector<double> x0, y0, z0;
    std::vector<int> face;
    std::vector<int> sgnX, sgnY, sgnZ;
};
#endif
