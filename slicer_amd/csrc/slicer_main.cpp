// slicer_main.cpp -- `SLICER_amd InputParams.ini [--devices 0-7]`: the Gadget branch of slicer-v2.cpp (:23-229) on the
// MI355X GPUs of one node, without MPI.  Planning (planner.cpp) -> plane loop -> createDensityMaps-equivalent passes
// over the C ABI -> writeMaps.  With several devices one host thread drives each GPU through its own handle: the
// sub-files of every snapshot are split over the devices in the reference's contiguous ranges (slicer-v2.cpp:162-175),
// the partial maps are summed onto device 0 in the accumulator type (slicer-v2.cpp:214-217 -> RCCL over xGMI,
// include/slicer_amd_rccl.h), and device 0's thread writes the FITS files -- byte-identical to a one-device run with
// --accum fixed64, within the f32 reorder bound otherwise.
// Differences from the reference driver, all opt-out:
//   * the planes cut from one box replication (same snapshot, same Random entry, same rcase) are built in ONE pass
//     over the snapshot (the reference re-reads and re-transforms it for each of them);   --single-plane disables
//   * nparttype* keys carry the real selected counts (the reference writes 0: densitymaps.cpp:497), which also makes
//     partinplanes runs write their per-type files;                                      --reference-counts disables
//   * SubFind / halo-catalogue mode (npix == 0) is not supported; with snopt > 0 and several devices every rank thread
//     draws from its own copy of the libc stream, like the reference's MPI ranks
//     (the thinning deviates come from the process-global libc rand() stream, densitymaps.cpp:387-397: the reference's
//     MPI ranks each own an identically seeded copy of it, host threads of one process would interleave their draws).
#include <dlfcn.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <valarray>
#include <vector>

#include "../../include/slicer_amd_rccl.h"
#include "fits_writer.hpp"
#include "gadget2_reader.hpp"
#include "planner.hpp"

using namespace slicer_amd;
using std::cerr;
using std::cout;
using std::endl;
using std::string;
using std::vector;

namespace {

bool file_exists(const string &p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}

string plane_label(int pll)  // slicer-v2.cpp:154-159
{
    char b[16];
    snprintf(b, sizeof b, "%i", pll);
    if (pll < 10)
        return string("00") + b;
    if (pll < 100)
        return string("0") + b;
    return b;
}

void dump_plan(const string &path, const InputParams &p, const Lens &lens, const Random &random,
               const vector<double> &snapbox, double fovradiants)
{
    FILE *f = fopen(path.c_str(), "w");
    if (!f)
        return;
    fprintf(f, "{\n \"Ds\": %.17g, \"fovradiants\": %.17g, \"nplanes\": %d, \"hydro\": %d,\n", p.Ds, fovradiants,
            lens.nplanes, (int)p.hydro);
    // the process's libc rand() stream as randomizeBox left it (31 words, oldest first): what shot-noise thinning
    // (snopt > 0) starts to draw from -- null where the C library does not expose it
    uint32_t v[31];
    if (slicer_libc_rand_state_get(v) == SLICER_OK) {
        fprintf(f, " \"libc_rand_state\": [");
        for (int i = 0; i < 31; i++)
            fprintf(f, "%u%s", v[i], i < 30 ? ", " : "],\n");
    } else {
        fprintf(f, " \"libc_rand_state\": null,\n");
    }
    fprintf(f, " \"planes\": [\n");
    for (int i = 0; i < lens.nplanes; i++) {
        fprintf(f,
                "  {\"ld\": %.17g, \"ld2\": %.17g, \"zsimlens\": %.17g, \"fromsnap\": \"%s\", \"fromsnapi\": %d, "
                "\"randomize\": %d, \"replication\": %d, \"nrepperp\": %d, \"snapbox\": %.17g, \"x0\": %.17g, \"y0\": %.17g, "
                "\"z0\": %.17g, \"face\": %d, \"sgn\": [%d, %d, %d]}%s\n",
                lens.ld[i], lens.ld2[i], lens.zsimlens[i], lens.fromsnap[i].c_str(), lens.fromsnapi[i],
                (int)lens.randomize[i], lens.replication[i], lens.nrepperp[i], snapbox[lens.fromsnapi[i]], random.x0[i],
                random.y0[i], random.z0[i], random.face[i], random.sgnX[i], random.sgnY[i], random.sgnZ[i],
                i + 1 < lens.nplanes ? "," : "");
    }
    fprintf(f, " ]\n}\n");
    fclose(f);
}

struct Span {
    SnapshotFile *snap;
    long base;
    const float *mass;
};
int fill_from_file(void *user, float *dst_pos, float *dst_mass, uint64_t first, uint64_t count)
{
    Span *s = static_cast<Span *>(user);
    if (!s->snap->read_at(s->base + (long)(12 * first), dst_pos, (size_t)(12 * count)))
        return 1;
    if (dst_mass)
        std::copy(s->mass + first, s->mass + first + count, dst_mass);
    return 0;
}

// "0-3", "0,2,5", "1": HIP device ordinals, one rank each
vector<int> parse_devices(const string &spec)
{
    vector<int> out;
    size_t i = 0;
    while (i < spec.size()) {
        size_t j = spec.find(',', i);
        if (j == string::npos)
            j = spec.size();
        const string tok = spec.substr(i, j - i);
        const size_t dash = tok.find('-');
        if (dash != string::npos && dash > 0) {
            for (int d = atoi(tok.substr(0, dash).c_str()); d <= atoi(tok.substr(dash + 1).c_str()); d++)
                out.push_back(d);
        } else if (!tok.empty()) {
            out.push_back(atoi(tok.c_str()));
        }
        i = j + 1;
    }
    return out;
}

// libslicer_amd_rccl.so is only needed (and only loaded) when more than one device takes part
struct RcclApi {
    void *lib = nullptr;
    int (*init_all)(slicer_rccl_comm *, int, const int *) = nullptr;
    int (*destroy)(slicer_rccl_comm) = nullptr;
    int (*plane_reduce)(slicer_handle, slicer_rccl_comm, int, int) = nullptr;  // slicer_rccl_plane_reduce_ex
    const char *(*last_error)(void) = nullptr;
    bool load()
    {
        lib = dlopen("libslicer_amd_rccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) {
            cerr << "slicer_amd: " << dlerror() << endl;
            return false;
        }
        init_all = (decltype(init_all))dlsym(lib, "slicer_rccl_comm_init_all");
        destroy = (decltype(destroy))dlsym(lib, "slicer_rccl_comm_destroy");
        plane_reduce = (decltype(plane_reduce))dlsym(lib, "slicer_rccl_plane_reduce_ex");
        last_error = (decltype(last_error))dlsym(lib, "slicer_rccl_last_error");
        return init_all && destroy && plane_reduce && last_error;
    }
};

// All rank threads meet here after their deposits and learn whether any of them failed: a collective is entered by
// every rank or by none (a rank that skipped it alone would leave the others waiting in RCCL for ever; the reference
// calls MPI_Abort in that situation, slicer-v2.cpp:204-207).
class Rendezvous {
    std::mutex m;
    std::condition_variable cv;
    int n, waiting = 0, generation = 0;
    bool failed = false, verdict = false;

public:
    explicit Rendezvous(int n_) : n(n_) {}
    bool any_failed(bool mine)  // blocks until all n ranks have called; the same answer for all of them
    {
        std::unique_lock<std::mutex> lk(m);
        failed = failed || mine;
        const int gen = generation;
        if (++waiting == n) {
            verdict = failed;
            failed = false;
            waiting = 0;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen; });
        }
        return verdict;
    }
};

// One rank = one GPU = one handle (+ its communicator).
struct Rank {
    int device = 0;
    slicer_handle h = nullptr;
    slicer_rccl_comm comm = nullptr;
    int rc = 0;
};

// The rank sum without RCCL (--reduce host): accumulators through host memory, summed in their own type onto rank 0.
// The same protocol as slicer_rccl_plane_reduce (include/slicer_amd.h "cross-rank sum"); it exists so that the
// multi-rank driver can be exercised where the ranks cannot form an RCCL clique (two handles on one GPU in the tests).
int host_plane_reduce(vector<Rank> &ranks, int npix, int n_planes)
{
    slicer_reduce_meta comb;
    for (int i = 0; i < SLICER_REDUCE_META_INTS; i++)
        comb.v[i] = INT32_MIN;
    for (auto &r : ranks) {
        slicer_reduce_meta m;
        if (slicer_reduce_meta_get(r.h, &m) != SLICER_OK) {
            cerr << "slicer_amd: " << slicer_last_error(r.h) << endl;
            return 1;
        }
        for (int i = 0; i < SLICER_REDUCE_META_INTS; i++)
            comb.v[i] = std::max(comb.v[i], m.v[i]);
    }
    for (auto &r : ranks)
        if (slicer_reduce_meta_set(r.h, &comb) != SLICER_OK) {
            cerr << "slicer_amd: " << slicer_last_error(r.h) << endl;
            return 1;
        }
    const size_t n = (size_t)npix * (size_t)npix;
    vector<unsigned char> sum, part;
    for (int p = 0; p < n_planes; p++) {
        void *acc0[7];
        int32_t elem = 0;
        if (slicer_plane_accumulators(ranks[0].h, p, acc0, &elem) != SLICER_OK)
            return 1;
        const size_t esz = elem == SLICER_ELEM_F32 ? 4 : 8;
        for (int s = 0; s < 7; s++) {
            if (!acc0[s])
                continue;
            sum.resize(n * esz);
            part.resize(n * esz);
            if (slicer_copy_to_host(ranks[0].h, sum.data(), acc0[s], n * esz) != SLICER_OK)
                return 1;
            for (size_t k = 1; k < ranks.size(); k++) {
                void *acc[7];
                int32_t e2 = 0;
                if (slicer_plane_accumulators(ranks[k].h, p, acc, &e2) != SLICER_OK || e2 != elem || !acc[s] ||
                    slicer_copy_to_host(ranks[k].h, part.data(), acc[s], n * esz) != SLICER_OK)
                    return 1;
                if (elem == SLICER_ELEM_F32) {
                    float *a = (float *)sum.data();
                    const float *b = (const float *)part.data();
                    for (size_t i = 0; i < n; i++)
                        a[i] = a[i] + b[i];
                } else if (elem == SLICER_ELEM_F64) {
                    double *a = (double *)sum.data();
                    const double *b = (const double *)part.data();
                    for (size_t i = 0; i < n; i++)
                        a[i] = a[i] + b[i];
                } else {
                    uint64_t *a = (uint64_t *)sum.data();
                    const uint64_t *b = (const uint64_t *)part.data();
                    for (size_t i = 0; i < n; i++)
                        a[i] += b[i];
                }
            }
            if (slicer_copy_to_device(ranks[0].h, acc0[s], sum.data(), n * esz) != SLICER_OK)
                return 1;
        }
        uint64_t *c0 = nullptr, tot[6], one[6];
        if (slicer_plane_device_counts(ranks[0].h, p, &c0) != SLICER_OK ||
            slicer_copy_to_host(ranks[0].h, tot, c0, sizeof tot) != SLICER_OK)
            return 1;
        for (size_t k = 1; k < ranks.size(); k++) {
            uint64_t *ck = nullptr;
            if (slicer_plane_device_counts(ranks[k].h, p, &ck) != SLICER_OK ||
                slicer_copy_to_host(ranks[k].h, one, ck, sizeof one) != SLICER_OK)
                return 1;
            for (int t = 0; t < 6; t++)
                tot[t] += one[t];
        }
        if (slicer_copy_to_device(ranks[0].h, c0, tot, sizeof tot) != SLICER_OK)
            return 1;
    }
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    string inifile, plan_path, devices_spec, reduce_mode = "rccl", reduce_algo = "rooted";
    int device = 0, mas = SLICER_MAS_TSC, accum = SLICER_ACC_F32;
    bool plan_only = false, single_plane = false, reference_counts = false, replication = false;
    for (int i = 1; i < argc; i++) {
        string a = argv[i];
        if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--devices" && i + 1 < argc) devices_spec = argv[++i];
        else if (a == "--reduce" && i + 1 < argc) reduce_mode = argv[++i];  // rccl (default) | host
        else if (a == "--reduce-algo" && i + 1 < argc) reduce_algo = argv[++i];  // rooted (default) | direct
        else if (a == "--ngp") mas = SLICER_MAS_NGP;
        else if (a == "--accum" && i + 1 < argc) {
            string v = argv[++i];
            accum = v == "f64" ? SLICER_ACC_F64 : (v == "fixed64" ? SLICER_ACC_FIXED64 : SLICER_ACC_F32);
        } else if (a == "--plan-only") plan_only = true;
        else if (a == "--dump-plan" && i + 1 < argc) plan_path = argv[++i];
        else if (a == "--single-plane") single_plane = true;
        else if (a == "--reference-counts") reference_counts = true;
        else if (a == "--replication") replication = true;  // -DUSE_REPLICATION (ReplicationOnPerpendicularPlane)
        else if (inifile.empty()) inifile = a;
        else {
            cerr << "unknown argument " << a << endl;
            return 2;
        }
    }
    if (inifile.empty()) {
        cout << "No params!! Nothing to be done!" << endl;  // slicer-v2.cpp:34
        return 2;
    }
    const int myid = 0;
    InputParams p{};
    double fovradiants = 0;
    if (readInput(p, inifile))
        return 1;
    if (p.simType == "SubFind") {
        cerr << "SubFind / halo light-cone mode (npix == 0) is outside this driver's scope" << endl;
        return 1;
    }
    vector<string> snappath;
    vector<double> snapred, snapbox;
    if (readRedList(p.filredshiftlist, snapred, snappath, snapbox, p))
        return 1;
    Header simdata{};
    {
        SnapshotFile s0;
        if (!s0.open(p.pathsnap + snappath[0] + ".0")) {
            cerr << "Error in opening the file: " << p.pathsnap + snappath[0] + ".0" << "!\n\a";
            return 1;
        }
        simdata = s0.header();
    }
    testHydro(p, simdata);

    // slicer-v2.cpp:79-96: distance table (h = 1) and the two interpolators
    const Cosmology cosmo{100.0, simdata.om0, simdata.oml, p.w, 0.0};
    vector<double> zl(kNeval);
    for (int i = 0; i < kNeval; i++)
        zl[i] = i * (p.zs + 1.0) / (kNeval - 1);
    vector<double> dl;
    try {
        dl = transverseDistanceTable(cosmo, zl);
    } catch (const std::exception &e) {
        cerr << e.what() << endl;
        return 1;
    }
    NaturalCubicSpline getDl, getZl;
    getDl.init(zl, dl);
    getZl.init(dl, zl);
    p.Ds = getDl.eval(p.zs);

    Lens lens{};
    if (buildPlanes(p, lens, snapred, snappath, snapbox, getDl, getZl, kNumberOfLensPerSnap, myid))
        return 1;
    lens.nrepperp.resize(lens.ld.size(), 0);
    for (size_t i = 0; i < lens.ld.size(); i++) {  // slicer-v2.cpp:103-125
        if (!replication) {
            if (testFov(p.fov, snapbox[lens.fromsnapi[i]] / 1e3 * kPosU, lens.ld2[i], myid, fovradiants))
                return 1;
        } else {
            computeReplications(p.fov, snapbox[lens.fromsnapi[i]] / 1e3 * kPosU, lens.ld2[i], myid, fovradiants,
                                lens.nrepperp[i]);
        }
    }
    Random random;
    randomizeBox(random, lens, p, kNumberOfLensPerSnap, myid);
    if (!plan_path.empty())
        dump_plan(plan_path, p, lens, random, snapbox, fovradiants);
    if (plan_only)
        return 0;
    // (snopt > 0: thinning consumes libc rand() plane by plane, densitymaps.cpp:387-397.  A pass over several planes
    // keeps that order -- the library replays its chunks plane-major when the pass ends -- so the planes of a box
    // replication still share one read of the snapshot.)

    vector<int> devs = devices_spec.empty() ? vector<int>{device} : parse_devices(devices_spec);
    if (devs.empty() || (reduce_mode != "rccl" && reduce_mode != "host") ||
        (reduce_algo != "rooted" && reduce_algo != "direct")) {
        cerr << "bad --devices / --reduce / --reduce-algo" << endl;
        return 2;
    }
    // snopt > 0 with several devices: the reference's MPI ranks each own an identically seeded copy of libc's rand()
    // stream (randomizeBox seeds it in every rank alike) and consume it independently (densitymaps.cpp:387-397).  The
    // rank threads of this process would interleave their draws on the one process-global stream, so every rank's
    // handle gets a stream of its own, started from the process state as randomizeBox left it -- the run then equals a
    // reference run on as many MPI ranks.  Where that state cannot be read (no glibc TYPE_3 generator) the combination
    // stays refused.
    // The same with ONE device: the stream is taken here, before the HIP runtime starts, and the handle thins from its
    // copy -- the runtime's own threads call rand() now and then (code-object loading at a kernel's first launch, ...),
    // which moves the process-global stream at unpredictable points of a run (round 3: planes of a cone differed from
    // run to run until the driver stopped drawing from the shared stream).
    uint32_t rank_stream[31];
    bool private_streams = p.snopt != 0;
    if (private_streams && slicer_libc_rand_state_get(rank_stream) != SLICER_OK) {
        if (devs.size() > 1) {
            cerr << "snopt > 0 on several devices needs per-rank copies of the libc rand() stream, which this C library "
                    "does not expose (slicer_libc_rand_supported() == 0): use a single device" << endl;
            return 2;
        }
        private_streams = false;  // one device: the process-global stream, drawn with rand() on the host
    }
    const int rccl_algo = reduce_algo == "direct" ? SLICER_RCCL_REDUCE_DIRECT : SLICER_RCCL_REDUCE_ROOTED;
    vector<Rank> ranks(devs.size());
    for (size_t r = 0; r < devs.size(); r++) {
        ranks[r].device = devs[r];
        if (slicer_create(devs[r], 1ull << 24, &ranks[r].h) != SLICER_OK) {
            cerr << "slicer_amd: " << slicer_last_error(nullptr) << endl;
            return 1;
        }
        if (private_streams && slicer_rand_stream_set(ranks[r].h, rank_stream) != SLICER_OK) {
            cerr << "slicer_amd: " << slicer_last_error(ranks[r].h) << endl;
            return 1;
        }
    }
    const int nranks = (int)ranks.size();
    RcclApi rccl;
    if (nranks > 1 && reduce_mode == "rccl") {
        vector<slicer_rccl_comm> comms(nranks);
        if (!rccl.load() || rccl.init_all(comms.data(), nranks, devs.data()) != SLICER_OK) {
            cerr << "slicer_amd: cannot set up RCCL over the devices: " << (rccl.last_error ? rccl.last_error() : "") << endl;
            return 1;
        }
        for (int r = 0; r < nranks; r++)
            ranks[r].comm = comms[r];
    }
    slicer_handle h = ranks[0].h;  // device 0 of the list is the root: it ends up with the sums and writes the maps
    Rendezvous rendezvous(nranks);
    cout << " Now loop on " << lens.nplanes << " planes " << endl;
    float rcase = 0.0f;  // slicer-v2.cpp:137
    int isnap = 0;
    int rc_all = 0;
    while (isnap < lens.nplanes && rc_all == 0) {
        // planes isnap .. iend-1 share snapshot, Random entry and rcase
        int iend = isnap + 1;
        if (!single_plane && !p.physical)
            while (iend < lens.nplanes && iend - isnap < SLICER_MAX_PLANES && !lens.randomize[iend] &&
                   lens.fromsnapi[iend] == lens.fromsnapi[isnap] && lens.nrepperp[iend] == lens.nrepperp[isnap])
                iend++;
        if (p.physical)  // slicer-v2.cpp:142-143
            p.npix = int((lens.ld2[isnap] + lens.ld[isnap]) / 2 * fovradiants / p.rgrid * 1e3 / kPosU) + 1;
        const string File = p.pathsnap + lens.fromsnap[isnap];
        if (lens.randomize[isnap])  // slicer-v2.cpp:184-185
            rcase = (float)(lens.ld[isnap] / snapbox[lens.fromsnapi[isnap]] * 1e3 / kPosU);
        for (int i = isnap + 1; i < iend; i++)
            if (lens.randomize[i])
                throw std::logic_error("plane grouping crossed a randomisation boundary");

        // resume: planes whose output exists are skipped (slicer-v2.cpp:188-202, only when !partinplanes)
        vector<int> todo;
        for (int i = isnap; i < iend; i++) {
            const string snappl = plane_label(lens.pll[i]);
            if (!p.partinplanes && file_exists(fileOutput(p, snappl))) {
                cout << fileOutput(p, snappl) << " Already exists" << endl;
                continue;
            }
            todo.push_back(i);
        }
        if (p.partinplanes && isnap == 0)
            cout << "!It is not possible to resume a Gadget run with partinplanes == true!" << endl;
        if (todo.empty()) {
            isnap = iend;
            continue;
        }
        SnapshotFile first;
        if (!first.open(File + ".0")) {
            cerr << "Error in opening the file: " << File + ".0" << "!\n\a";
            rc_all = 1;
            break;
        }
        const Header simhdr = first.header();
        first.close();

        slicer_plane_desc d{};
        d.npix = p.npix;
        d.n_planes = (int)todo.size();
        d.mas = mas;
        d.accum = accum;
        d.hydro = p.hydro;
        d.snopt = p.snopt;
        d.want_type_maps = p.partinplanes ? 1 : 0;
        d.fov_rad = fovradiants;
        for (size_t k = 0; k < todo.size(); k++) {
            d.ld[k] = lens.ld[todo[k]];
            d.ld2[k] = lens.ld2[todo[k]];
            d.nrepperp[k] = lens.nrepperp[todo[k]];
        }
        // one rank's share of the pass: its contiguous range of sub-files (slicer-v2.cpp:162-175: numfiles / nranks each,
        // the last rank takes the remainder), then the rank sum
        auto deposit_rank = [&](int r) {
            Rank &R = ranks[r];
            slicer_handle hr = R.h;
            R.rc = 0;
            auto failed = [&](const char *what) {
                cerr << "slicer_amd (device " << R.device << "): " << what << ": " << slicer_last_error(hr) << endl;
                R.rc = 1;
            };
            if (slicer_plane_begin(hr, &d) != SLICER_OK)
                return failed("plane_begin");
            const int per = simhdr.numfiles / nranks;
            const int ffmin = r * per, ffmax = (r == nranks - 1) ? simhdr.numfiles : (r + 1) * per;
            for (int ff = ffmin; ff < ffmax && R.rc == 0; ff++) {
                char suffix[32];
                snprintf(suffix, sizeof suffix, "%i", ff);
                SnapshotFile snap;
                if (!snap.open(File + "." + suffix)) {
                    cerr << "Error in opening the file: " << File << "." << suffix << "!\n\a";
                    R.rc = 1;
                    break;
                }
                const Header &data = snap.header();
                long pos_off = 0, pos_bytes = 0;
                vector<float> mass[6];
                if (!snap.locate_block("POS ", pos_off, pos_bytes) || (p.hydro && !snap.read_masses(mass))) {
                    cerr << "slicer_amd: cannot read POS / MASS of " << snap.path() << endl;
                    R.rc = 1;
                    break;
                }
                slicer_file_desc f{};
                for (int t = 0; t < 6; t++) {
                    f.npart[t] = data.npart[t];
                    f.massarr[t] = data.massarr[t];
                }
                f.boxsize = data.boxsize;
                f.sgn[0] = random.sgnX[isnap];
                f.sgn[1] = random.sgnY[isnap];
                f.sgn[2] = random.sgnZ[isnap];
                f.face = random.face[isnap];
                f.center[0] = random.x0[isnap];
                f.center[1] = random.y0[isnap];
                f.center[2] = random.z0[isnap];
                f.rcase = rcase;
                if (slicer_file_begin(hr, &f) != SLICER_OK)
                    return failed("file_begin");
                size_t off = 0;
                for (int t = 0; t < 6 && R.rc == 0; t++) {
                    const size_t n = data.npart[t] > 0 ? (size_t)data.npart[t] : 0;
                    if (n) {
                        const float *m = (p.hydro && data.massarr[t] == 0 && !mass[t].empty()) ? mass[t].data() : nullptr;
                        Span span{&snap, pos_off + (long)(12 * off), m};
                        if (slicer_deposit_stream(hr, t, n, m != nullptr, fill_from_file, &span) != SLICER_OK)
                            return failed("deposit");
                    }
                    off += n;
                }
                if (R.rc == 0 && slicer_file_end(hr) != SLICER_OK)
                    return failed("file_end");
            }
        };
        auto run_rank = [&](int r) {
            deposit_rank(r);
            Rank &R = ranks[r];
            // slicer-v2.cpp:214-217: the sum over ranks onto the root, here over RCCL on the accumulators.  The ranks
            // agree on the outcome of the deposit phase first: the collective is entered by all of them or by none.
            if (nranks > 1 && rendezvous.any_failed(R.rc != 0)) {
                if (R.rc == 0)
                    cerr << "slicer_amd (device " << R.device << "): another rank failed; skipping the rank sum" << endl;
                R.rc = 1;
                return;
            }
            if (R.rc == 0 && nranks > 1 && R.comm && rccl.plane_reduce(R.h, R.comm, 0, rccl_algo) != SLICER_OK) {
                cerr << "slicer_amd (device " << R.device << "): rank sum: " << rccl.last_error() << endl;
                R.rc = 1;
            }
        };
        {
            vector<std::thread> threads;
            for (int r = 1; r < nranks; r++)
                threads.emplace_back(run_rank, r);
            run_rank(0);
            for (auto &t : threads)
                t.join();
            for (auto &R : ranks)
                rc_all |= R.rc;
            if (rc_all == 0 && nranks > 1 && reduce_mode == "host")
                rc_all = host_plane_reduce(ranks, p.npix, (int)todo.size());
        }
        if (rc_all)
            break;
        const size_t np2 = (size_t)p.npix * (size_t)p.npix;
        std::valarray<float> tot(np2), toti[6];
        for (size_t k = 0; k < todo.size() && rc_all == 0; k++) {
            const int i = todo[k];
            int64_t nsel[6];
            float *d_toti[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            if (slicer_plane_read(h, (int)k, &tot[0], nullptr, nsel) != SLICER_OK ||
                (p.partinplanes && slicer_plane_device_maps(h, (int)k, nullptr, d_toti) != SLICER_OK)) {
                cerr << "slicer_amd: " << slicer_last_error(h) << endl;
                rc_all = 1;
                break;
            }
            int ntotxyi[6];
            for (int t = 0; t < 6 && rc_all == 0; t++) {
                ntotxyi[t] = reference_counts ? 0 : (int)nsel[t];
                if (!p.partinplanes)
                    continue;
                // per-type maps straight from the device into the array writeMaps gets; types without particles: zeros
                if (toti[t].size() != np2)
                    toti[t].resize(np2);
                if (d_toti[t]) {
                    if (slicer_copy_to_host(h, &toti[t][0], d_toti[t], np2 * sizeof(float)) != SLICER_OK) {
                        cerr << "slicer_amd: " << slicer_last_error(h) << endl;
                        rc_all = 1;
                    }
                } else {
                    toti[t] = 0.0f;
                }
            }
            if (rc_all)
                break;
            const double zsim = getZl.eval((lens.ld2[i] + lens.ld[i]) / 2.0);  // slicer-v2.cpp:219
            Header hd = simhdr;
            try {
                writeMaps(p, hd, lens, i, zsim, plane_label(lens.pll[i]), p.snpix, tot, toti, ntotxyi, myid);
            } catch (const std::runtime_error &) {
                rc_all = 1;
            }
        }
        isnap = iend;
    }
    for (auto &R : ranks) {
        if (R.comm)
            rccl.destroy(R.comm);
        slicer_destroy(R.h);
    }
    return rc_all;
}
