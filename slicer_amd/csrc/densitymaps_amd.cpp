// densitymaps_amd.cpp -- createDensityMaps (densitymaps.cpp:419-524) over the C ABI of include/slicer_amd.h.
// Host orchestration only; every particle goes through libslicer_amd.so's HIP kernels.
#include "densitymaps_amd.hpp"

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>
#include <vector>

#include "../../include/slicer_amd.h"
#include "gadget2_reader.hpp"

namespace {

// The planes the last pass produced besides the one it was asked for.  slicer-v2.cpp calls createDensityMaps once per
// lens plane, and the planes of one box replication (lens.randomize set on the first only) read the same sub-files with
// the same Random entry and rcase: the adapter deposits them all in the pass of the first call (one file read, one H2D
// copy, one run of the kernels for up to SLICER_MAX_PLANES planes) and leaves the others on the device, where the
// following calls pick them up.  Every entry serves one call; any other call starts a fresh pass.
struct PlaneGroup {
    bool valid = false;
    int first = 0, n = 0;
    bool used[SLICER_MAX_PLANES] = {};
    std::string file;
    unsigned ffmin = 0, ffmax = 0;
    double fov = 0;
    float rcase = 0;
    int npix = 0, hydro = 0, mas = 0, accum = 0, skip_types = 0;
};

struct AdapterState {
    slicer_handle h = nullptr;
    int device = -1;
    int mas = SLICER_MAS_TSC, accum = SLICER_ACC_F32, algo = SLICER_ALGO_AUTO, true_counts = 0;
    int want_device = -1;
    int skip_type_maps = -1;  // -1: SLICER_AMD_SKIP_TYPE_MAPS from the environment (default 0)
    PlaneGroup grp;
};
AdapterState g;

int env_int(const char *k, int dflt)
{
    const char *v = getenv(k);
    return v ? atoi(v) : dflt;
}

bool ensure_handle(int myid)
{
    if (g.h)
        return true;
    int dev = g.want_device >= 0 ? g.want_device : env_int("SLICER_AMD_DEVICE", -1);
    if (dev < 0) {
        int ndev = env_int("SLICER_AMD_NUM_DEVICES", 1);  // one MPI rank per GPU: rank -> device
        dev = ndev > 0 ? myid % ndev : 0;
    }
    // particles per staged chunk: file reads of chunk k+1 overlap the H2D copy and kernels of chunk k, so a sub-file
    // should span several chunks (2^22: 10.8 ms per 2^24-particle sub-file end to end, against 12.5 ms at 2^24)
    const uint64_t chunk = (uint64_t)env_int("SLICER_AMD_CHUNK_LOG2", 22);
    // Shot-noise thinning (snopt > 0, densitymaps.cpp:387-397) draws from libc's rand() stream as randomizeBox left it.
    // This is the first call of the run (slicer-v2.cpp: randomizeBox, then the plane loop), and the last moment at which
    // that stream is still what the reference would see: the HIP runtime's own threads call rand() now and then once it
    // runs (a kernel's first launch loads its code object, ...).  So the stream is read BEFORE the runtime starts and the
    // handle thins from its own copy (slicer_rand_stream_set); the process's stream stays where randomizeBox left it.
    uint32_t stream[31];
    const bool have_stream = slicer_libc_rand_state_get(stream) == SLICER_OK;
    int rc = slicer_create(dev, 1ull << chunk, &g.h);
    if (rc != SLICER_OK) {
        std::cerr << "slicer_amd: " << slicer_last_error(nullptr) << std::endl;
        g.h = nullptr;
        return false;
    }
    if (have_stream)
        (void)slicer_rand_stream_set(g.h, stream);
    g.device = dev;
    g.mas = env_int("SLICER_AMD_NGP", g.mas == SLICER_MAS_NGP) ? SLICER_MAS_NGP : SLICER_MAS_TSC;
    return true;
}

}  // namespace

extern "C" void slicer_amd_adapter_config(int mas, int accum, int algo, int true_counts, int device)
{
    g.mas = mas;
    g.accum = accum;
    g.algo = algo;
    g.true_counts = true_counts;
    g.want_device = device;
}

extern "C" void slicer_amd_adapter_skip_type_maps(int on) { g.skip_type_maps = on ? 1 : 0; }

extern "C" void slicer_amd_adapter_shutdown(void)
{
    if (g.h)
        slicer_destroy(g.h);
    g.h = nullptr;
    g.grp.valid = false;
}

int createDensityMaps(InputParams &p, Lens &lens, Random &random, int isnap, unsigned int ffmin, unsigned int ffmax,
                      std::string File, double fovradiants, double rcase, gsl_spline *, gsl_interp_accel *,
                      gsl_spline *, gsl_interp_accel *, std::valarray<float> &mapxytot,
                      std::valarray<float> (&mapxytoti)[6], int (&ntotxyi)[6], int myid)
{
    const size_t np2 = (size_t)p.npix * (size_t)p.npix;
    // densitymaps.cpp:426-431 resizes (= zero-fills) all seven maps.  Every map is either overwritten from the device
    // or zero-filled below, so arrays that already have the right size are not touched twice (7 x 64 MiB at 4096^2).
    if (mapxytot.size() != np2)
        mapxytot.resize(np2);
    bool stale[6];  // per-type arrays that keep the previous call's contents and must read zero unless overwritten
    for (int i = 0; i < 6; i++) {
        ntotxyi[i] = 0;
        stale[i] = mapxytoti[i].size() == np2;
        if (!stale[i])
            mapxytoti[i].resize(np2);
    }
    // The reference zero-fills all seven maps on entry.  The six per-type arrays (6 x 64 MiB at 4096^2) are cleared by
    // a helper thread while the files are read and the GPU works; the populated ones are overwritten from the device
    // at the end.
    struct Zeroer {
        std::thread t;
        ~Zeroer()
        {
            if (t.joinable())
                t.join();
        }
    } zeroer;
    zeroer.t = std::thread([&mapxytoti, &stale, np2]() {
        for (int i = 0; i < 6; i++)
            if (stale[i])
                memset(&mapxytoti[i][0], 0, np2 * sizeof(float));
    });
    if (!ensure_handle(myid))
        return 1;
    slicer_handle h = g.h;
    // Opt-in (slicer_amd_adapter_skip_type_maps / SLICER_AMD_SKIP_TYPE_MAPS=1): without partinplanes the caller never
    // reads mapxytoti (writeMaps, densitymaps.cpp:537-584, only writes the all-types map), so the per-type maps are
    // neither built on the device nor copied back -- they come back zero-filled; mapxytot is unchanged in the TSC
    // accumulator's tolerance (one shared accumulator instead of the f32 sum of six) and bitwise under NGP.
    if (g.skip_type_maps < 0)
        g.skip_type_maps = env_int("SLICER_AMD_SKIP_TYPE_MAPS", 0) ? 1 : 0;
    const int skip_types = (g.skip_type_maps && !p.partinplanes) ? 1 : 0;

    // a plane left on the device by the pass of an earlier call (see PlaneGroup)?
    PlaneGroup &G = g.grp;
    int slot = 0;
    bool made_group = false;
    const bool hit = G.valid && isnap > G.first && isnap < G.first + G.n && !G.used[isnap - G.first] && G.file == File &&
                     G.ffmin == ffmin && G.ffmax == ffmax && G.fov == fovradiants && G.rcase == (float)rcase &&
                     G.npix == p.npix && G.hydro == (p.hydro ? 1 : 0) && G.mas == g.mas && G.accum == g.accum &&
                     G.skip_types == skip_types;
    if (hit) {
        slot = isnap - G.first;
        G.used[slot] = true;
    } else {
        G.valid = false;
        // planes that follow in the same box replication: same snapshot, same Random entry, no new randomisation
        // (slicer-v2.cpp:184-185 keeps rcase for them).  SLICER_AMD_LOOKAHEAD=0 restores one plane per pass.
        int n = 1;
        auto same = [&](int j) {
            const size_t k = (size_t)j, i = (size_t)isnap;
            return k < lens.randomize.size() && k < lens.fromsnap.size() && k < lens.ld.size() && k < lens.ld2.size() &&
                   k < lens.nrepperp.size() && k < random.x0.size() && k < random.face.size() && !lens.randomize[k] &&
                   lens.fromsnap[k] == lens.fromsnap[i] && random.x0[k] == random.x0[i] && random.y0[k] == random.y0[i] &&
                   random.z0[k] == random.z0[i] && random.face[k] == random.face[i] && random.sgnX[k] == random.sgnX[i] &&
                   random.sgnY[k] == random.sgnY[i] && random.sgnZ[k] == random.sgnZ[i];
        };
        if (env_int("SLICER_AMD_LOOKAHEAD", 1) && p.snopt == 0 && !p.physical && isnap < (int)lens.fromsnap.size())
            while (n < SLICER_MAX_PLANES && isnap + n < lens.nplanes && same(isnap + n))
                n++;
        slicer_plane_desc d{};
        d.npix = p.npix;
        d.n_planes = n;
        d.mas = g.mas;
        d.accum = g.accum;
        d.algo = g.algo;
        d.hydro = p.hydro ? 1 : 0;
        d.snopt = p.snopt;
        d.want_type_maps = skip_types ? 0 : 1;
        d.fov_rad = fovradiants;
        for (int j = 0; j < n; j++) {
            d.ld[j] = lens.ld[isnap + j];
            d.ld2[j] = lens.ld2[isnap + j];
            d.nrepperp[j] = lens.nrepperp[isnap + j];
        }
        if (slicer_plane_begin(h, &d) != SLICER_OK) {
            std::cerr << "slicer_amd: " << slicer_last_error(h) << std::endl;
            return 1;
        }
        if (n > 1) {
            made_group = true;
            G.first = isnap;
            G.n = n;
            for (int j = 0; j < SLICER_MAX_PLANES; j++)
                G.used[j] = j == 0;
            G.file = File;
            G.ffmin = ffmin;
            G.ffmax = ffmax;
            G.fov = fovradiants;
            G.rcase = (float)rcase;
            G.npix = p.npix;
            G.hydro = p.hydro ? 1 : 0;
            G.mas = g.mas;
            G.accum = g.accum;
            G.skip_types = skip_types;
        }
    }

    for (unsigned int ff = ffmin; ff < ffmax && !hit; ff++) {
        char suffix[32];
        snprintf(suffix, sizeof suffix, "%i", (int)ff);  // sconv(ff, fINT)
        const std::string file_in = File + "." + suffix;
        slicer_amd::SnapshotFile snap;
        if (!snap.open(file_in)) {
            std::cerr << "Error in opening the file: " << file_in << "!\n\a";  // gadget2io.cpp:20
            return 1;
        }
        const Header &data = snap.header();
        long pos_off = 0, pos_bytes = 0;
        if (!snap.locate_block("POS ", pos_off, pos_bytes)) {
            std::cerr << "slicer_amd: no POS block in " << snap.path() << std::endl;
            return 1;
        }
        std::vector<float> mass[6];
        if (p.hydro && !snap.read_masses(mass)) {
            std::cerr << "slicer_amd: cannot read MASS/BHMA in " << snap.path() << std::endl;
            return 1;
        }
        slicer_file_desc f{};
        size_t ntot = 0;
        for (int t = 0; t < 6; t++) {
            f.npart[t] = data.npart[t];
            f.massarr[t] = data.massarr[t];
            ntot += data.npart[t] > 0 ? (size_t)data.npart[t] : 0;
        }
        if ((size_t)pos_bytes < 12 * ntot) {
            std::cerr << "slicer_amd: POS block of " << snap.path() << " is shorter than the header says" << std::endl;
            return 1;
        }
        f.boxsize = data.boxsize;
        f.sgn[0] = random.sgnX[isnap];
        f.sgn[1] = random.sgnY[isnap];
        f.sgn[2] = random.sgnZ[isnap];
        f.face = random.face[isnap];
        f.center[0] = random.x0[isnap];
        f.center[1] = random.y0[isnap];
        f.center[2] = random.z0[isnap];
        f.rcase = (float)rcase;  // readPos takes "float rcase" (gadget2io.h:122)
        if (slicer_file_begin(h, &f) != SLICER_OK) {
            std::cerr << "slicer_amd: " << slicer_last_error(h) << std::endl;
            return 1;
        }
        // The POS block is streamed straight into the library's pinned staging buffers (no pageable copy of the
        // block): file reads overlap with the H2D copies and kernels of the previous chunk.
        struct Span {
            slicer_amd::SnapshotFile *snap;
            long base;          // file offset of this type's first particle
            const float *mass;  // or nullptr
        };
        auto fill = [](void *user, float *dst_pos, float *dst_mass, uint64_t first, uint64_t count) -> int {
            Span *s = static_cast<Span *>(user);
            if (!s->snap->read_at(s->base + (long)(12 * first), dst_pos, (size_t)(12 * count)))
                return 1;
            if (dst_mass)
                std::copy(s->mass + first, s->mass + first + count, dst_mass);
            return 0;
        };
        size_t off = 0;
        for (int t = 0; t < 6; t++) {
            const size_t n = data.npart[t] > 0 ? (size_t)data.npart[t] : 0;
            if (n) {
                const float *m = (p.hydro && data.massarr[t] == 0 && !mass[t].empty()) ? mass[t].data() : nullptr;
                Span span{&snap, pos_off + (long)(12 * off), m};
                if (slicer_deposit_stream(h, t, n, m != nullptr, fill, &span) != SLICER_OK) {
                    std::cerr << "slicer_amd: " << slicer_last_error(h) << std::endl;
                    return 1;
                }
            }
            off += n;
        }
        if (slicer_file_end(h) != SLICER_OK) {
            std::cerr << "slicer_amd: " << slicer_last_error(h) << std::endl;
            return 1;
        }
        if (myid == 0)
            std::cout << " done map*tot " << std::endl;
    }

    // all-types map and counters, then every per-type map straight from the device into the caller's array (types
    // that never appeared have no device map: zeros)
    int64_t nsel[6] = {0, 0, 0, 0, 0, 0};
    int rc = slicer_plane_read(h, slot, &mapxytot[0], nullptr, nsel);
    if (rc != SLICER_OK) {
        std::cerr << "slicer_amd: " << slicer_last_error(h) << std::endl;
        if (rc == SLICER_ERR_NEGATIVE_COORD)
            std::cerr << "Aborting from Rank " << myid << std::endl;  // densitymaps.cpp:343
        return 1;
    }
    float *d_toti[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (!skip_types && slicer_plane_device_maps(h, slot, nullptr, d_toti) != SLICER_OK) {
        std::cerr << "slicer_amd: " << slicer_last_error(h) << std::endl;
        return 1;
    }
    if (zeroer.t.joinable())
        zeroer.t.join();  // every per-type array now reads zero
    for (int i = 0; i < 6; i++) {
        if (d_toti[i] && slicer_copy_to_host(h, &mapxytoti[i][0], d_toti[i], np2 * sizeof(float)) != SLICER_OK) {
            std::cerr << "slicer_amd: " << slicer_last_error(h) << std::endl;
            return 1;
        }
        ntotxyi[i] = g.true_counts ? (int)nsel[i] : 0;
    }
    if (made_group)
        G.valid = true;  // the pass went through: its other planes wait on the device
    if (myid == 0)
        std::cout << " maps done! from Rank:" << myid << std::endl;
    return 0;
}
