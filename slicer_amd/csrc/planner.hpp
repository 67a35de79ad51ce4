// planner.hpp -- host-side planning of the light cone: the O(nplanes) work that produces
// the hot path's inputs.  Restates, function by function and with the same names, the reference's
//   readInput        data.cpp:8-87            readRedList   gadget2io.cpp:613-661   testHydro gadget2io.cpp:34-48
//   w0waCDM          w0waCDM.{h,cpp}          getSnap / buildPlanes / randomizeBox / testFov / computeReplications
//                                             densitymaps.cpp:9-283
// The two GSL csplines of slicer-v2.cpp:88-94 become NaturalCubicSpline (GSL is not installed here, so the last
// bits of interpolated values -- zsimlens, the REDSHIFT key -- are UNPINNED against GSL; plane edges ld/ld2, the
// snapshot choice and the randomisation do not depend on them except at exact ties).  randomizeBox draws from
// libc's srand/rand exactly as the reference does, so with the same glibc it yields the same Random plan.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "slicer_types.hpp"

namespace slicer_amd {

constexpr double kPosU = 1.0;            // gadget2io.h:14
constexpr int kNumberOfLensPerSnap = 4;  // densitymaps.h:23
constexpr int kNeval = 1000;             // slicer-v2.cpp:5

class w0waCDM {  // w0waCDM.h:20-60
public:
    w0waCDM(double H0, double OmegaM, double OmegaLambda, double w0, double wa);
    double comovingDistance(double z) const;
    double transverseComovingDistance(double z) const;

private:
    static constexpr double CSPEEDOFLIGHT = 2.99792458e+3 * 100;
    double H0, OmegaM, OmegaLambda, w0, wa;
    mutable std::map<double, double> cache;
    double Hz(double z) const;
};

class NaturalCubicSpline {  // stands where gsl_interp_cspline + gsl_spline_eval stand in the reference
public:
    void init(const std::vector<double> &x, const std::vector<double> &y);
    double eval(double x) const;

private:
    std::vector<double> x_, y_, c_;
};

int readInput(InputParams &p, const std::string &name);
int readRedList(const std::string &filredshiftlist, std::vector<double> &snapred, std::vector<std::string> &snappath,
                std::vector<double> &snapbox, InputParams &p);
void testHydro(InputParams &p, const Header &data);
int getSnap(std::vector<double> &zsnap, const NaturalCubicSpline &GetDl, double dlens);
int buildPlanes(InputParams &p, Lens &lens, std::vector<double> &snapred, std::vector<std::string> &snappath,
                std::vector<double> &snapbox, const NaturalCubicSpline &GetDl, const NaturalCubicSpline &GetZl,
                int numOfLensPerSnap, int myid);
void randomizeBox(Random &random, Lens &lens, InputParams &p, int numOfLensPerSnap, int myid,
                  bool fixed_plc_vertex = false);
int testFov(double fov, double boxl, double Ds, int myid, double &fovradiants);
void computeReplications(double fov, double boxl, double Ds, int myid, double &fovradiants, int &nrepperp);

}  // namespace slicer_amd
