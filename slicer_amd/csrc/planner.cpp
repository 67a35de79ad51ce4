#include "planner.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>

#include "gadget2_reader.hpp"

namespace slicer_amd {

using std::cerr;
using std::cout;
using std::endl;
using std::string;
using std::vector;

// ---------------------------------------------------------------- distance table
// What the reference's start-up needs from its cosmology class (w0waCDM.{h,cpp}, used at slicer-v2.cpp:79-86) is one
// table: the transverse comoving distance at an increasing grid of redshifts starting at 0.  Numeric contract kept,
// because the plane edges derive from it: each grid interval is integrated with the trapezoid rule in steps of 1/100 of
// the interval, the abscissa advancing by repeated addition until it reaches the interval's end -- so rounding lets some
// intervals take a 101st trapezoid (a +0.06 % bias of the reference that a "better" quadrature would not reproduce).
double expansionRate(const Cosmology &c, double z)
{
    const double a1 = 1.0 + z;
    const double dark = c.omegaLambda * pow(a1, 3.0 * (1.0 + c.w0 + c.wa)) * exp(-3.0 * c.wa * z / a1);
    const double curvature = (1.0 - c.omegaM - c.omegaLambda) * pow(a1, 2);
    return c.h0 * sqrt(dark + c.omegaM * pow(a1, 3) + curvature);
}

std::vector<double> transverseDistanceTable(const Cosmology &c, const std::vector<double> &zgrid)
{
    if (!(c.h0 > 0) || c.omegaM < 0 || c.omegaLambda < 0)
        throw std::invalid_argument("cosmology: H0 must be positive and the density parameters non-negative");
    constexpr double kC = 2.99792458e+3 * 100;  // km/s
    constexpr int kStepsPerInterval = 100;
    std::vector<double> out(zgrid.size(), 0.0);
    double integral = 0.0;  // int_0^z dz' / H(z')
    for (size_t i = 0; i < zgrid.size(); i++) {
        if (i > 0) {
            const double from = zgrid[i - 1], to = zgrid[i], h = (to - from) / kStepsPerInterval;
            for (double z = from; z < to; z += h)
                integral += 0.5 * h * (1.0 / expansionRate(c, z) + 1.0 / expansionRate(c, z + h));
        }
        const double radial = integral * kC;  // Mpc (h = 1 units when H0 = 100)
        const double omegaK = 1.0 - c.omegaM - c.omegaLambda;
        if (fabs(omegaK) < 1e-5) {
            out[i] = radial;
        } else {
            // curved: the reference's expression in its own operation order (w0waCDM.cpp:75,79 -- including its choice
            // of sinh for Omega_K < 0), so that the roundings, and with them the plane edges, are the same
            const double rootK = sqrt(fabs(omegaK));
            out[i] = omegaK < 0 ? kC / c.h0 / rootK * sinh(rootK * c.h0 / kC * radial)
                                : kC / c.h0 / rootK * sin(rootK * c.h0 / kC * radial);
        }
    }
    return out;
}

// ---------------------------------------------------------------- natural cubic spline (GSL cspline's definition:
// second derivative zero at both ends; y(x) = y_i + b_i dx + c_i dx^2 + d_i dx^3 on [x_i, x_{i+1}])
void NaturalCubicSpline::init(const vector<double> &x, const vector<double> &y)
{
    const size_t n = x.size();
    x_ = x;
    y_ = y;
    c_.assign(n, 0.0);
    if (n < 3)
        return;
    const size_t m = n - 2;  // interior unknowns c_1 .. c_{n-2}
    vector<double> diag(m), off(m), rhs(m);
    for (size_t i = 0; i < m; i++) {
        const double h_i = x[i + 1] - x[i], h_ip1 = x[i + 2] - x[i + 1];
        const double ydiff_i = y[i + 1] - y[i], ydiff_ip1 = y[i + 2] - y[i + 1];
        off[i] = h_ip1;
        diag[i] = 2.0 * (h_ip1 + h_i);
        rhs[i] = 3.0 * (ydiff_ip1 / h_ip1 - ydiff_i / h_i);
    }
    // Thomas algorithm on the symmetric tridiagonal system
    vector<double> cp(m), dp(m);
    cp[0] = off[0] / diag[0];
    dp[0] = rhs[0] / diag[0];
    for (size_t i = 1; i < m; i++) {
        const double den = diag[i] - off[i - 1] * cp[i - 1];
        cp[i] = off[i] / den;
        dp[i] = (rhs[i] - off[i - 1] * dp[i - 1]) / den;
    }
    c_[m] = dp[m - 1];
    for (size_t i = m - 1; i-- > 0;)
        c_[i + 1] = dp[i] - cp[i] * c_[i + 2];
}

double NaturalCubicSpline::eval(double x) const
{
    const size_t n = x_.size();
    if (n < 2)
        return n ? y_[0] : 0.0;
    size_t lo = 0, hi = n - 1;  // bisection for the interval, clamped to the table like gsl_interp_bsearch
    while (hi > lo + 1) {
        const size_t mid = (lo + hi) / 2;
        if (x_[mid] > x)
            hi = mid;
        else
            lo = mid;
    }
    const double dx = x_[lo + 1] - x_[lo], dy = y_[lo + 1] - y_[lo];
    const double b = dy / dx - dx * (c_[lo + 1] + 2.0 * c_[lo]) / 3.0;
    const double d = (c_[lo + 1] - c_[lo]) / (3.0 * dx);
    const double t = x - x_[lo];
    return y_[lo] + t * (b + t * (c_[lo] + t * d));
}

// ---------------------------------------------------------------- data.cpp:8-87
int readInput(InputParams &p, const string &name)
{
    std::ifstream fin(name.c_str());
    if (!fin.is_open()) {
        cerr << "slicer_amd: cannot open the parameter file '" << name << "' (path relative to the working directory)"
             << endl;
        return 1;  // data.cpp:15-18 stops the program here; this driver returns the error instead
    }
    string str;
    auto next = [&](string &dst) {
        std::getline(fin, str);
        std::getline(fin, dst);
    };
    try {
        next(str); p.npix = std::stoi(str);
        next(str); p.zs = std::stof(str);
        next(str); p.fov = std::stof(str);
        next(p.filredshiftlist);
        next(p.pathsnap);
        next(p.simulation);
        next(str); p.seedcenter = std::stoi(str);
        next(str); p.seedface = std::stoi(str);
        next(str); p.seedsign = std::stoi(str);
        next(str); p.partinplanes = std::stoi(str);
        next(p.directory);
        next(p.suffix);
        next(str); p.snopt = std::stoi(str);
        next(str); p.w = std::stof(str);
    } catch (const std::exception &) {
        cerr << " Params file " << name << " is malformed" << endl;
        return 1;
    }
    p.simType = (p.npix == 0) ? "SubFind" : "Gadget";
    p.physical = (p.npix < 0);
    p.hydro = false;
    p.rgrid = 0;
    char buf[32];
    if (!p.physical) {
        snprintf(buf, sizeof buf, "%i", p.npix);
        p.snpix = buf;
    } else {
        int n = -p.npix;
        snprintf(buf, sizeof buf, "%i", n);
        p.snpix = string(buf) + "_kpc";
        p.rgrid = n;
    }
    if (p.snopt < 0) {
        cerr << "Impossible value for Shot-Noise option!" << endl;
        return 1;
    }
    return 0;
}

// ---------------------------------------------------------------- snapshot list (behaviour of gadget2io.cpp:613-661)
// The list file names one snapshot per entry, sorted by redshift.  Entries are taken until one reaches the source
// redshift (it is still included) or the file ends; every entry's first sub-file must open; |z| < 1e-5 counts as 0.
int readRedList(const string &filredshiftlist, vector<double> &snapred, vector<string> &snappath,
                vector<double> &snapbox, InputParams &p)
{
    std::ifstream list(filredshiftlist.c_str());
    if (!list.is_open()) {
        cerr << " snapshot list " << filredshiftlist << " cannot be opened: stopping" << endl;
        return 1;
    }
    double previous = -999.9;
    for (;;) {
        string name;
        list >> name;  // (at end of file this leaves an empty name, which then fails to open: the reference's behaviour)
        snappath.push_back(name);
        const string first = p.pathsnap + name + ".0";
        SnapshotFile snap;
        if (!snap.open(first)) {
            cerr << "Error in opening the file: " << first << "!\n\a";
            cerr << name << " not found!" << endl;
            return 1;
        }
        const Header &hdr = snap.header();
        if (hdr.redshift < previous) {
            cerr << " Snapshots on " << filredshiftlist << " are not sorted!" << endl;
            return 1;
        }
        previous = std::abs(hdr.redshift) < 1e-5 ? 0.0 : hdr.redshift;
        snapred.push_back(previous);
        snapbox.push_back(hdr.boxsize);
        if (!(hdr.redshift < p.zs) || list.eof())
            break;
    }
    return 0;
}

// ---------------------------------------------------------------- hydro run? (behaviour of gadget2io.cpp:34-48)
// A Gadget snapshot carries per-particle masses iff some species is present whose table mass is zero.
void testHydro(InputParams &p, const Header &data)
{
    if (p.simType != "Gadget")
        return;
    bool per_particle = false;
    for (int t = 0; t < 6; t++)
        per_particle = per_particle || (data.npart[t] != 0 && data.massarr[t] == 0);
    p.hydro = per_particle;
}

// ---------------------------------------------------------------- densitymaps.cpp:9-33
// Index of the snapshot nearest to `dlens` in comoving distance (-1 for an empty list).  Behaviour kept from the
// reference: the distance gap is rounded to binary32 before it is compared, the first of equal gaps wins, and a list
// whose every snapshot is 99999 Mpc/h or further away answers 0.
int getSnap(vector<double> &zsnap, const NaturalCubicSpline &GetDl, double dlens)
{
    int nearest = zsnap.empty() ? -1 : 0;
    double smallest = 99999;
    for (int k = 0; k < (int)zsnap.size(); k++) {
        const float gap = (float)fabs(GetDl.eval(zsnap[k]) - dlens);
        if (gap < smallest) {
            smallest = gap;
            nearest = k;
        }
    }
    return nearest;
}

// ---------------------------------------------------------------- densitymaps.cpp:46-156
// Plane grid: slabs of (box / numOfLensPerSnap) piled from the observer until the source distance p.Ds is passed.
// For every slab the snapshot is the one whose redshift is nearest to the redshift of the slab; switching to a
// snapshot with a different box size is only allowed at a replication boundary (every numOfLensPerSnap slabs).
int buildPlanes(InputParams &p, Lens &lens, vector<double> &snapred, vector<string> &snappath, vector<double> &snapbox,
                const NaturalCubicSpline &GetDl, const NaturalCubicSpline &GetZl, int numOfLensPerSnap, int myid)
{
    const size_t nsnaps = snapred.size();
    auto slab_depth = [&](size_t snap) { return snapbox[snap] / (1e3 / kPosU) / numOfLensPerSnap; };

    int in_use = 0;          // snapshot the previous slab was cut from
    int slabs_in_run = 0;    // slabs since the snapshot last changed
    int n_slabs = 0;
    double far_edge = 0.0;   // comoving distance of the last slab's far edge (Mpc/h)
    do {
        n_slabs++;
        slabs_in_run++;
        const bool at_replication_boundary = n_slabs == 1 || (n_slabs - 1) % numOfLensPerSnap == 0;

        // trial: which snapshot suits a slab ending at far_edge + its own depth?
        double best_dz = 9999;
        int choice = in_use;
        for (size_t cand = (size_t)in_use; cand < nsnaps; cand++) {
            const double trial_edge = far_edge + slab_depth(cand);
            const int nearest = getSnap(snapred, GetDl, trial_edge);
            if (nearest == -1) {
                cerr << "snapred is an empty array!" << endl;
                cerr << "Check your snapshot list file." << endl;
                return 1;
            }
            if ((size_t)nearest >= snapred.size()) {
                cerr << "getSnap returned an index outside the range! " << endl;
                return 1;
            }
            const double dz = fabs(snapred[nearest] - GetZl.eval(trial_edge));
            const bool may_switch = at_replication_boundary || snapbox[nearest] == snapbox[in_use];
            if (dz < best_dz && may_switch) {
                choice = nearest;
                best_dz = dz;
            }
        }
        const double depth = slab_depth((size_t)choice);
        far_edge += depth;
        const double centre = far_edge - 0.5 * depth;
        const double z_centre = GetZl.eval(centre);
        const int snap = getSnap(snapred, GetDl, centre);  // the slab is finally cut from the snapshot nearest its centre
        if (myid == 0)
            cout << " simulation snapshots = " << far_edge << "  " << GetZl.eval(far_edge) << "  " << n_slabs
                 << " from snap " << snappath[snap] << "  " << z_centre << endl;
        lens.ld.push_back(far_edge - slab_depth((size_t)snap));
        lens.ld2.push_back(far_edge);
        lens.zfromsnap.push_back(snapred[snap]);
        if (n_slabs != 1 && snap != in_use) {  // close the run of the previous snapshot
            lens.replication.insert(lens.replication.end(), (size_t)(slabs_in_run - 1), n_slabs - 1);
            slabs_in_run = 1;
        }
        in_use = snap;
        lens.zsimlens.push_back(z_centre);
        lens.fromsnap.push_back(snappath[in_use]);
        lens.fromsnapi.push_back(in_use);
        lens.randomize.push_back(at_replication_boundary);
    } while (far_edge < p.Ds);
    // the last run (the reference pushes one entry more than it has slabs; .back() is what is read)
    lens.replication.insert(lens.replication.end(), (size_t)(slabs_in_run + 1), n_slabs);

    if (myid == 0) {
        cout << " Comoving Distance of the last plane " << p.Ds << endl;
        cout << " nsnaps = " << nsnaps << "\n" << endl;
    }
    std::ofstream planelist;
    if (myid == 0)
        planelist.open((p.directory + "planes_list_" + p.suffix + ".txt").c_str());
    for (size_t i = 0; i < lens.fromsnap.size(); i++) {
        if (myid == 0) {
            cout << lens.zsimlens[i] << " planes = " << lens.ld[i] << "  " << lens.ld2[i] << "  " << lens.replication[i]
                 << " from snap " << lens.fromsnap[i] << endl;
            // columns read back by Lens/kslicer.py: index, z, Dl low, Dl up, replication, snapshot, z snapshot, randomize
            planelist << i << "   " << lens.zsimlens[i] << "   " << lens.ld[i] << "   " << lens.ld2[i] << "   "
                      << lens.replication[i] << "   " << lens.fromsnap[i] << "   " << lens.zfromsnap[i] << "  "
                      << lens.randomize[i] << endl;
        }
        lens.pll.push_back((int)i);
    }
    lens.nplanes = lens.replication.back();
    return 0;
}

// ---------------------------------------------------------------- densitymaps.cpp:166-248
// One Random entry per plane; drawn afresh at replication boundaries (lens.randomize), copied otherwise.  The
// three seeds are re-armed per replication g = i / numOfLensPerSnap with strides 13, 5 and 8, and every draw is
// rand()/float(RAND_MAX): the same libc calls in the same order as the reference, hence the same plan.
void randomizeBox(Random &random, Lens &lens, InputParams &p, int numOfLensPerSnap, int myid, bool fixed_plc_vertex)
{
    const size_t n = (size_t)lens.replication.back();
    for (auto *v : {&random.x0, &random.y0, &random.z0})
        v->resize(n);
    for (auto *v : {&random.sgnX, &random.sgnY, &random.sgnZ, &random.face})
        v->resize(n);
    auto uniform01 = []() { return rand() / float(RAND_MAX); };
    auto draw_sign = [&]() {
        int s = 2;
        while (s > 1 || s < 0)
            s = int(uniform01() + 0.5);
        return s == 0 ? -1 : s;
    };
    for (size_t i = 0; i < n; i++) {
        if (!lens.randomize[i]) {  // same box replication as the previous plane
            random.x0[i] = random.x0[i - 1];
            random.y0[i] = random.y0[i - 1];
            random.z0[i] = random.z0[i - 1];
            random.face[i] = random.face[i - 1];
            random.sgnX[i] = random.sgnX[i - 1];
            random.sgnY[i] = random.sgnY[i - 1];
            random.sgnZ[i] = random.sgnZ[i - 1];
        } else {
            const size_t g = i / numOfLensPerSnap;
            srand(p.seedcenter + g * 13);
            if (fixed_plc_vertex) {  // -DUSE_FIXED_PLC_VERTEX: observer on the box axis
                random.x0[i] = 0.0;
                random.y0[i] = 0.0;
                random.z0[i] = 0.5;
            } else {
                random.x0[i] = uniform01();
                random.y0[i] = uniform01();
                random.z0[i] = uniform01();
            }
            srand(p.seedface + g * 5);
            int face = 7;
            while (face > 6 || face < 1)
                face = int(1 + uniform01() * 5. + 0.5);
            random.face[i] = face;
            srand(p.seedsign + g * 8);
            random.sgnX[i] = draw_sign();
            random.sgnY[i] = draw_sign();
            random.sgnZ[i] = draw_sign();
        }
        if (myid == 0) {
            cout << "  " << endl;
            cout << " random centers  for the box " << i << " = " << random.x0[i] << "  " << random.y0[i] << "  "
                 << random.z0[i] << endl;
            cout << " face of the dice " << random.face[i] << endl;
            cout << " signs of the coordinates = " << random.sgnX[i] << "  " << random.sgnY[i] << " " << random.sgnZ[i]
                 << endl;
        }
    }
}

// ---------------------------------------------------------------- field of view vs box (behaviour of densitymaps.cpp:255-283)
// The field of view at the far edge of a plane must fit the box (no lateral replication), or -- with
// -DUSE_REPLICATION -- the number of lateral box copies needed on each side is worked out.
static double degToRad(double deg) { return deg / 180. * M_PI; }

int testFov(double fov, double boxl, double Ds, int myid, double &fovradiants)
{
    fovradiants = degToRad(fov);
    const bool fits = !(fovradiants * Ds > boxl);
    if (fits || myid != 0)  // only rank 0 reports (and stops): the reference's behaviour
        return 0;
    // ("Field view too large" is the phrase the reference prints: kept for whoever greps the logs)
    cerr << " !!Field view too large!! " << fov << " deg does not fit the box at comoving distance " << Ds
         << " (at most " << boxl / Ds * 180. / M_PI << " deg): stopping" << endl;
    return 1;
}

void computeReplications(double fov, double boxl, double Ds, int, double &fovradiants, int &nrepperp)
{
    fovradiants = degToRad(fov);
    const double half_width = Ds * tan(fovradiants / 2.0);  // half extent of the field at distance Ds
    const double beyond = half_width - boxl / 2.0;          // what sticks out of one box on each side
    nrepperp = beyond <= 0 ? 0 : (int)ceil(beyond / boxl);
}

}  // namespace slicer_amd
