// adapter_driver.cpp -- calls createDensityMaps the way slicer-v2.cpp:138-229 does (one plane), on snapshot
// files prepared by the pytest that launches it, and dumps the maps for comparison with the oracle.
// usage: adapter_driver <File base> <ffmin> <ffmax> <npix> <fov_rad> <ld> <ld2> <rcase> <ngp> <hydro> <out.bin>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <valarray>

#include "densitymaps_amd.hpp"

int main(int argc, char **argv)
{
    if (argc != 12) {
        fprintf(stderr, "bad usage\n");
        return 2;
    }
    InputParams p{};
    p.npix = atoi(argv[4]);
    p.hydro = atoi(argv[10]) != 0;
    p.simType = "Gadget";
    p.snopt = 0;
    p.partinplanes = true;
    Lens lens{};
    lens.nplanes = 1;
    lens.ld = {atof(argv[6])};
    lens.ld2 = {atof(argv[7])};
    lens.nrepperp = {0};
    Random random{};
    random.x0 = {0.3};
    random.y0 = {0.6};
    random.z0 = {0.1};
    random.face = {3};
    random.sgnX = {-1};
    random.sgnY = {1};
    random.sgnZ = {-1};
    slicer_amd_adapter_config(atoi(argv[9]) ? 1 : 0, 0, 0, 0, 0);
    std::valarray<float> mapxytot, mapxytoti[6];
    int ntotxyi[6];
    // ADAPTER_REPEAT=N: call N times and print every call's wall time (the first includes the device context)
    const int repeat = getenv("ADAPTER_REPEAT") ? std::max(1, atoi(getenv("ADAPTER_REPEAT"))) : 1;
    int rc = 0;
    for (int r = 0; r < repeat && rc == 0; r++) {
        const auto t0 = std::chrono::steady_clock::now();
        rc = createDensityMaps(p, lens, random, 0, (unsigned)atoi(argv[2]), (unsigned)atoi(argv[3]), argv[1],
                               atof(argv[5]), atof(argv[8]), nullptr, nullptr, nullptr, nullptr, mapxytot, mapxytoti,
                               ntotxyi, 1);
        if (repeat > 1)
            fprintf(stderr, "createDensityMaps call %d: %.1f ms\n", r,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    slicer_amd_adapter_shutdown();
    if (rc)
        return 1;
    std::ofstream out(argv[11], std::ios::binary);
    out.write((const char *)&mapxytot[0], sizeof(float) * mapxytot.size());
    for (int i = 0; i < 6; i++)
        out.write((const char *)&mapxytoti[i][0], sizeof(float) * mapxytoti[i].size());
    out.write((const char *)ntotxyi, sizeof ntotxyi);
    return 0;
}
