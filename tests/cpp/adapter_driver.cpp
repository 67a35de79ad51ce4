// adapter_driver.cpp -- calls createDensityMaps the way slicer-v2.cpp:138-229 does (plane by plane), on snapshot
// files prepared by the pytest that launches it, and dumps the maps for comparison with the oracle.
// usage: adapter_driver <File base> <ffmin> <ffmax> <npix> <fov_rad> <ld> <ld2> <rcase> <ngp> <hydro> <out.bin>
// <ld> and <ld2> may be comma-separated lists: the planes of one box replication (randomize set on the first only,
// one Random entry), one createDensityMaps call each, in order; the output holds every plane's 7 maps + 6 counters.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>
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
    // ADAPTER_SNOPT=k: shot-noise thinning; ADAPTER_SRAND=seed: the libc stream as randomizeBox would leave it (srand +
    // five draws), set before the first createDensityMaps call -- i.e. before the HIP runtime starts
    p.snopt = getenv("ADAPTER_SNOPT") ? atoi(getenv("ADAPTER_SNOPT")) : 0;
    if (getenv("ADAPTER_SRAND")) {
        srand((unsigned)atoi(getenv("ADAPTER_SRAND")));
        for (int i = 0; i < 5; i++)
            (void)rand();
    }
    p.partinplanes = getenv("ADAPTER_PARTINPLANES") ? atoi(getenv("ADAPTER_PARTINPLANES")) != 0 : true;
    auto split = [](const char *a) {
        std::vector<double> v;
        std::string t(a);
        size_t at = 0;
        while (at <= t.size()) {
            const size_t c = t.find(',', at);
            v.push_back(atof(t.substr(at, c == std::string::npos ? c : c - at).c_str()));
            if (c == std::string::npos)
                break;
            at = c + 1;
        }
        return v;
    };
    Lens lens{};
    lens.ld = split(argv[6]);
    lens.ld2 = split(argv[7]);
    const int nplanes = (int)lens.ld.size();
    if ((int)lens.ld2.size() != nplanes) {
        fprintf(stderr, "ld / ld2 lists differ in length\n");
        return 2;
    }
    lens.nplanes = nplanes;
    lens.nrepperp.assign(nplanes, 0);
    lens.fromsnap.assign(nplanes, "snap");
    lens.fromsnapi.assign(nplanes, 0);
    lens.randomize.assign(nplanes, false);
    lens.randomize[0] = true;
    Random random{};
    random.x0.assign(nplanes, 0.3);
    random.y0.assign(nplanes, 0.6);
    random.z0.assign(nplanes, 0.1);
    random.face.assign(nplanes, 3);
    random.sgnX.assign(nplanes, -1);
    random.sgnY.assign(nplanes, 1);
    random.sgnZ.assign(nplanes, -1);
    slicer_amd_adapter_config(atoi(argv[9]) ? 1 : 0, 0, 0, 0, 0);
    std::valarray<float> mapxytot, mapxytoti[6];
    int ntotxyi[6];
    std::ofstream out;
    // ADAPTER_REPEAT=N: run the plane loop N times and print every call's wall time (the first includes the device
    // context)
    const int repeat = getenv("ADAPTER_REPEAT") ? std::max(1, atoi(getenv("ADAPTER_REPEAT"))) : 1;
    int rc = 0;
    for (int r = 0; r < repeat && rc == 0; r++)
        for (int isnap = 0; isnap < nplanes && rc == 0; isnap++) {
            const auto t0 = std::chrono::steady_clock::now();
            rc = createDensityMaps(p, lens, random, isnap, (unsigned)atoi(argv[2]), (unsigned)atoi(argv[3]), argv[1],
                                   atof(argv[5]), atof(argv[8]), nullptr, nullptr, nullptr, nullptr, mapxytot, mapxytoti,
                                   ntotxyi, 1);
            if (repeat > 1 || getenv("ADAPTER_TIMES"))
                fprintf(stderr, "createDensityMaps call %d plane %d: %.1f ms\n", r, isnap,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            if (rc == 0 && r == repeat - 1) {
                if (!out.is_open())
                    out.open(argv[11], std::ios::binary);
                out.write((const char *)&mapxytot[0], sizeof(float) * mapxytot.size());
                for (int i = 0; i < 6; i++)
                    out.write((const char *)&mapxytoti[i][0], sizeof(float) * mapxytoti[i].size());
                out.write((const char *)ntotxyi, sizeof ntotxyi);
            }
        }
    slicer_amd_adapter_shutdown();
    return rc ? 1 : 0;
}
