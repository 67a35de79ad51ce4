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

// ---------------------------------------------------------------- w0waCDM.cpp:8-84
w0waCDM::w0waCDM(double H0_, double OmegaM_, double OmegaLambda_, double w0_, double wa_)
    : H0(H0_), OmegaM(OmegaM_), OmegaLambda(OmegaLambda_), w0(w0_), wa(wa_)
{
    if (H0 <= 0 || OmegaM < 0 || OmegaLambda < 0)
        throw std::invalid_argument("Invalid cosmological parameters: H0 must be positive, and density parameters "
                                    "cannot be negative.");
}

double w0waCDM::Hz(double z) const
{
    double rhoLambda = OmegaLambda * pow(1 + z, 3 * (1 + w0 + wa)) * exp(-3 * wa * z / (1 + z));
    double rhoM = OmegaM * pow(1 + z, 3);
    double rhoTot = rhoLambda + rhoM + (1 - OmegaM - OmegaLambda) * pow(1 + z, 2);
    return H0 * sqrt(rhoTot);
}

double w0waCDM::comovingDistance(double z) const
{
    // NB (reference behaviour kept): a cache hit returns the stored value WITHOUT the c factor (w0waCDM.cpp:33-36);
    // the driver only ever asks for increasing, distinct redshifts, so the hit path is not taken there.
    if (cache.find(z) != cache.end())
        return cache[z];
    double distance = 0;
    double lastZ = 0;
    double dz = 1e-4;
    auto it = cache.lower_bound(z);
    if (it != cache.begin()) {
        --it;
        distance = it->second;
        lastZ = it->first;
        dz = (z - lastZ) / 100;
    }
    for (double zi = lastZ; zi < z; zi += dz)  // trapezoids; the float-accumulated loop bound is the reference's
        distance += 0.5 * dz * (1.0 / Hz(zi) + 1.0 / Hz(zi + dz));
    cache[z] = distance;
    return cache[z] * CSPEEDOFLIGHT;
}

double w0waCDM::transverseComovingDistance(double z) const
{
    double D_C = comovingDistance(z);
    if (fabs(1 - OmegaM - OmegaLambda) < 1e-5)
        return D_C;
    double OmegaK = 1.0 - OmegaM - OmegaLambda;
    double sqrtOmegaK = sqrt(fabs(OmegaK));
    if (OmegaK < 0)
        return CSPEEDOFLIGHT / H0 / sqrtOmegaK * sinh(sqrtOmegaK * H0 / CSPEEDOFLIGHT * D_C);
    return CSPEEDOFLIGHT / H0 / sqrtOmegaK * sin(sqrtOmegaK * H0 / CSPEEDOFLIGHT * D_C);
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
        cerr << " Params file " << name << " does not exist where you are running the code " << endl;
        cerr << " I will STOP here!!! " << endl;
        return 1;  // the reference calls exit(1) here
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

// ---------------------------------------------------------------- gadget2io.cpp:613-661
int readRedList(const string &filredshiftlist, vector<double> &snapred, vector<string> &snappath,
                vector<double> &snapbox, InputParams &p)
{
    std::ifstream redlist(filredshiftlist.c_str());
    double zlast = -999.9;
    if (!redlist.is_open()) {
        cerr << " redshift list file redshift_list.txt does not " << endl;
        cerr << " exist in the Code dir ... check this out      " << endl;
        cerr << "    I will STOP here !!! " << endl;
        return 1;
    }
    Header header{};
    do {
        string name;
        redlist >> name;
        snappath.push_back(name);
        SnapshotFile snap;
        if (!snap.open(p.pathsnap + name + ".0")) {
            cerr << "Error in opening the file: " << p.pathsnap + name + ".0" << "!\n\a";
            cerr << name << " not found!" << endl;
            return 1;
        }
        header = snap.header();
        if (header.redshift < zlast) {
            cerr << " Snapshots on " << filredshiftlist << " are not sorted!" << endl;
            return 1;
        }
        zlast = header.redshift;
        if (std::abs(zlast) < 1e-5)
            zlast = 0.0;
        snapred.push_back(zlast);
        snapbox.push_back(header.boxsize);
    } while ((header.redshift < p.zs) & (!redlist.eof()));
    return 0;
}

// ---------------------------------------------------------------- gadget2io.cpp:34-48
void testHydro(InputParams &p, const Header &data)
{
    if (p.simType.compare("Gadget") == 0) {
        int dimmass0 = 0;
        for (int i = 0; i <= 5; i++)
            if (data.massarr[i] == 0)
                dimmass0 += data.npart[i];
        p.hydro = bool(dimmass0);
    }
}

// ---------------------------------------------------------------- densitymaps.cpp:9-33
int getSnap(vector<double> &zsnap, const NaturalCubicSpline &GetDl, double dlens)
{
    if (zsnap.empty())
        return -1;
    unsigned int pos = 0;
    double aux = 99999;
    for (size_t i = 0; i < zsnap.size(); i++) {
        float test = (float)std::abs(GetDl.eval(zsnap[i]) - dlens);  // the reference stores |.| in a float
        if (test < aux) {
            aux = test;
            pos = (unsigned)i;
        }
    }
    return (int)pos;
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

// ---------------------------------------------------------------- densitymaps.cpp:255-283
int testFov(double fov, double boxl, double Ds, int myid, double &fovradiants)
{
    fovradiants = fov / 180. * M_PI;
    if ((fovradiants)*Ds > boxl && myid == 0) {
        cerr << " !!Field view too large!!\n !!!I will STOP here!!! " << endl;
        cerr << " Value set is = " << fov << endl;
        cerr << " Maximum value allowed " << boxl / Ds * 180. / M_PI << " in degrees " << endl;
        cerr << " For the lens at " << Ds << endl;
        return 1;
    }
    return 0;
}

void computeReplications(double fov, double boxl, double Ds, int, double &fovradiants, int &nrepperp)
{
    fovradiants = fov / 180. * M_PI;
    if (Ds * tan(fovradiants / 2.0) <= boxl / 2.0)
        nrepperp = 0;
    else
        nrepperp = (int)ceil((Ds * tan(fovradiants / 2) - boxl / 2.0) / boxl);
}

}  // namespace slicer_amd
