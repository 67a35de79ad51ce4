// planner.hpp -- host-side planning of the light cone: the O(nplanes) work that produces
// the hot path's inputs.  Restates, function by function and with the same names, the reference's
//   readInput        data.cpp:8-87            readRedList   gadget2io.cpp:613-661   testHydro gadget2io.cpp:34-48
//   distance table   w0waCDM.{h,cpp} as used at slicer-v2.cpp:79-86 (transverseDistanceTable, own structure)
//   getSnap / buildPlanes / randomizeBox / testFov / computeReplications      densitymaps.cpp:9-283
// The two GSL csplines of slicer-v2.cpp:88-94 become NaturalCubicSpline (GSL is not installed here, so the last
// bits of interpolated values -- zsimlens, the REDSHIFT key -- are UNPINNED against GSL; plane edges ld/ld2, the
// snapshot choice and the randomisation do not depend on them except at exact ties).  randomizeBox draws from
// libc's srand/rand exactly as the reference does, so with the same glibc it yields the same Random plan.
#pragma once
#include <string>
#include <vector>

#include "slicer_types.hpp"

namespace slicer_amd {

constexpr double kPosU = 1.0;            // gadget2io.h:14
constexpr int kNumberOfLensPerSnap = 4;  // densitymaps.h:23
constexpr int kNeval = 1000;             // slicer-v2.cpp:5

// Background cosmology of the distance table (the reference keeps these in its w0waCDM class, w0waCDM.h:20-60).
struct Cosmology {
    double h0;           // km/s/Mpc (the driver uses 100: distances in Mpc/h)
    double omegaM, omegaLambda;
    double w0, wa;       // dark-energy equation of state w(a) = w0 + wa (1 - a)
};
double expansionRate(const Cosmology &c, double z);
// Transverse comoving distances at an increasing redshift grid that starts at 0 (what slicer-v2.cpp:79-86 tabulates),
// with the reference's integration contract; throws std::invalid_argument on unphysical parameters.
std::vector<double> transverseDistanceTable(const Cosmology &c, const std::vector<double> &zgrid);

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
