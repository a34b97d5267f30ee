// crd_run.cpp -- command-line driver with the reference's surface: `<program> <ini file>` writes the per-subdomain
// text files the reference's Python plotting / torus-mapping utilities read.  It replaces main() of the four
// reference programs (src/FHNmodel_torus.cpp:148-497 and siblings); invoked through one of the alias names
// FHNmodel_torus / FHNmodel_flat / GoldbeterModel_torus / GoldbeterModel_flat it takes exactly one argument, like
// they do.  Time integration is fixed-step RK4 on the GPU (libcrd) by default; --adaptive runs the reference's integrator,
// ARKode's default explicit pair and controller restated (CRD_ADAPT_ARKODE), --adaptive-rk43 the RK4(3) pair of earlier rounds.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "crd.h"

namespace {

struct Options {
	int model = -1, surface = -1;
	int gpus = 0;        // 0 = take [Solver] gpus from the ini (default 1)
	int devices = 0;     // number of physical devices to spread the slabs over (0 = as many as slabs)
	double dt = -1.0;
	int stepper = -1;
	int precision = -1;
	std::string outdir = ".";
	std::string ini;
	bool quiet = false;
	int adaptive = -1;
	bool binary = false;       // also write <Model>_<surface>_<var>.NNN.npy (crd_npy_writer)
	bool binary_only = false;  // ... and no text rows (the text files are created empty; only crdmodel_amd.post reads such a run)
	bool ref_steady_state = false;  // Goldbeter rest state as the reference reads it from its script's print (8 decimals)
	int d0 = 0, d1 = 0;  // --decomp d0xd1: the reference's 2-D block layout (theta x phi) instead of phi-slabs; "--decomp mpi" = MPI_Dims_create(gpus)
	bool block_contexts = false;  // --block-contexts: also COMPUTE on those blocks (staged kernels); default: files in that layout, computed on phi-slabs
	bool decomp_mpi = false;
};

[[noreturn]] void usage(const char *argv0, bool alias)
{
	if (alias) {
		std::cerr << "Usage: " << argv0 << " <Config file path>";  // src/FHNmodel_torus.cpp:153
	} else {
		std::cerr << "Usage: " << argv0
		          << " --model fhn|goldbeter --surface torus|flat [--gpus G] [--devices D] [--dt DT] [--stepper auto|staged|fused]\n"
		             "       [--precision 64|32] [--adaptive|--adaptive-rk43|--fixed] [--binary|--binary-only] [--ref-steady-state] [--decomp D0xD1|mpi [--block-contexts]]\n"
		             "       [--outdir DIR] [--quiet]\n"
		             "       <Config file path>\n";
	}
	std::exit(EXIT_FAILURE);
}

bool preset_from_name(const std::string &base, Options *o)
{
	if (base == "FHNmodel_torus") { o->model = CRD_MODEL_FHN; o->surface = CRD_SURFACE_TORUS; return true; }
	if (base == "FHNmodel_flat") { o->model = CRD_MODEL_FHN; o->surface = CRD_SURFACE_FLAT; return true; }
	if (base == "GoldbeterModel_torus") { o->model = CRD_MODEL_GOLDBETER; o->surface = CRD_SURFACE_TORUS; return true; }
	if (base == "GoldbeterModel_flat") { o->model = CRD_MODEL_GOLDBETER; o->surface = CRD_SURFACE_FLAT; return true; }
	return false;
}

void banner(const crd_run_config &cfg, const crd_grid &g, int n_slabs, int64_t nxl0, int64_t nyl0, double s0, double s1, double dt, int64_t steps_per_output)
{
	// Same lines as src/FHNmodel_torus.cpp:249-275 (and the Goldbeter variant, src/GoldbeterModel_torus.cpp:266-303),
	// with rtol / atol replaced by the fixed step.
	const crd_params &p = cfg.params;
	const bool fhn = p.model == CRD_MODEL_FHN, torus = p.surface == CRD_SURFACE_TORUS;
	std::cout << (fhn ? "\n2D FHN model PDE problem on a " : "\n Goldbeter model PDE problem on a ") << (torus ? "torus" : "flat surface") << ":\n";
	std::cout << "   nprocs = " << n_slabs << "\n";
	std::cout << "   nx = " << g.nx << "\n";
	std::cout << "   ny = " << g.ny << "\n";
	std::cout << "   nxl = " << nxl0 << "\n";
	std::cout << "   nyl = " << nyl0 << "\n";
	std::cout << "   Diff = " << p.diffusion << "\n";
	std::cout << "   Tfinal = " << cfg.t_final << "\n";
	std::cout << "   Output timesteps = " << cfg.output_timestep << "\n";
	if (torus) {
		std::cout << "   Major circumference = " << p.surface_length << "\n";
		std::cout << "   Minor circumference = " << p.surface_width << "\n";
	} else {
		std::cout << "   Surface length = " << p.surface_length << "\n";
		std::cout << "   Surface width = " << p.surface_width << "\n";
	}
	if (fhn) std::cout << "   Absorbing boundary turn off time = " << p.t_boundary << "\n";
	std::cout << "   Wavelength = " << cfg.wave_length * 100 << "%\n";
	std::cout << "   Wavewidth = " << cfg.wave_width * 100 << "%\n";
	if (fhn && torus) std::cout << "   Wave inside = " << cfg.wave_inside << "\n";
	if (cfg.adaptive == 2) std::cout << "   integrator = adaptive RK4(3) on GPU\n   rtol = " << cfg.rtol << "\n   atol = " << cfg.atol << "\n";
	else if (cfg.adaptive) std::cout << "   integrator = ARKode-style ERK on GPU (Zonneveld 5(3)4, PID controller)\n   rtol = " << cfg.rtol << "\n   atol = " << cfg.atol << "\n";
	else std::cout << "   integrator = classical RK4 on GPU, dt = " << dt << " (" << steps_per_output << " steps per output)\n";
	if (!fhn && p.just_diffusion == 1) {
		std::cout << "   Diffusion Only\n\n";
		return;
	}
	std::cout << "   Include all variables in output = " << cfg.include_all_vars << "\n";
	if (!fhn) std::cout << "   Absorbing boundary turn off time = " << p.t_boundary << "\n";
	if (p.vary_beta == 0) {
		std::cout << "   Beta = " << p.beta << "\n";
		std::cout << "   Stable state values: " << (fhn ? "U = " : "Z = ") << s0 << (fhn ? ", V = " : ", Y = ") << s1 << "\n\n";
	} else {
		std::cout << "   Beta varied over " << (torus ? "torus" : "surface") << "\n";
		if (!fhn) {
			if (cfg.ic_type == 0) std::cout << "   Homogeneous ICs\n";
			if (cfg.ic_type == 1) std::cout << "   ICs: initial perturbation\n";
			if (cfg.ic_type == 2) std::cout << "   Random ICs\n";
		}
		std::cout << "\n";
	}
}

int die(const char *what, int rc, crd_ctx *ctx)
{
	std::cerr << "\nCRD_ERROR: " << what << " failed with flag = " << rc << " (" << crd_status_string(rc) << ")";
	const char *detail = crd_last_error(ctx);
	if (detail && *detail) std::cerr << ": " << detail;
	std::cerr << "\n\n";
	return 1;
}

}  // namespace

// Rank and size an MPI launcher gave this process through its environment (no MPI library is linked).
struct Launcher {
	int rank = 0, size = 1;
	const char *kind = "";
};
static Launcher detect_launcher()
{
	static const struct {
		const char *rank, *size, *kind;
	} known[] = {{"OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE", "Open MPI"},
	             {"PMI_RANK", "PMI_SIZE", "MPICH / Hydra"},
	             {"MV2_COMM_WORLD_RANK", "MV2_COMM_WORLD_SIZE", "MVAPICH"}};
	Launcher l;
	for (const auto &k : known) {
		const char *r = std::getenv(k.rank), *n = std::getenv(k.size);
		if (r && n && std::atoi(n) >= 1 && std::atoi(r) >= 0 && std::atoi(r) < std::atoi(n)) {
			l.rank = std::atoi(r);
			l.size = std::atoi(n);
			l.kind = k.kind;
			break;
		}
	}
	return l;
}

int main(int argc, char *argv[])
{
	Options o;
	std::string base = argv[0];
	const size_t slash = base.find_last_of('/');
	if (slash != std::string::npos) base = base.substr(slash + 1);
	const bool alias = preset_from_name(base, &o);

	if (alias) {
		if (argc != 2) usage(argv[0], true);  // src/FHNmodel_torus.cpp:151-155
		o.ini = argv[1];
	} else {
		for (int a = 1; a < argc; a++) {
			const std::string s = argv[a];
			auto next = [&]() -> std::string {
				if (a + 1 >= argc) usage(argv[0], false);
				return argv[++a];
			};
			if (s == "--model") {
				const std::string v = next();
				o.model = v == "fhn" ? CRD_MODEL_FHN : v == "goldbeter" ? CRD_MODEL_GOLDBETER : -1;
			} else if (s == "--surface") {
				const std::string v = next();
				o.surface = v == "torus" ? CRD_SURFACE_TORUS : v == "flat" ? CRD_SURFACE_FLAT : -1;
			} else if (s == "--gpus") o.gpus = std::atoi(next().c_str());
			else if (s == "--devices") o.devices = std::atoi(next().c_str());
			else if (s == "--dt") o.dt = std::atof(next().c_str());
			else if (s == "--outdir") o.outdir = next();
			else if (s == "--quiet") o.quiet = true;
			else if (s == "--adaptive") o.adaptive = 1;
			else if (s == "--adaptive-rk43") o.adaptive = 2;
			else if (s == "--fixed") o.adaptive = 0;
			else if (s == "--binary") o.binary = true;
			else if (s == "--binary-only") o.binary = o.binary_only = true;
			else if (s == "--ref-steady-state") o.ref_steady_state = true;
			else if (s == "--block-contexts") o.block_contexts = true;
			else if (s == "--decomp") {
				const std::string v = next();
				if (v == "mpi") o.decomp_mpi = true;
				else if (std::sscanf(v.c_str(), "%dx%d", &o.d0, &o.d1) != 2 || o.d0 < 1 || o.d1 < 1) usage(argv[0], false);
			}
			else if (s == "--precision") {
				const std::string v = next();
				o.precision = v == "32" ? CRD_PRECISION_F32 : v == "64" ? CRD_PRECISION_F64 : -2;
			} else if (s == "--stepper") {
				const std::string v = next();
				o.stepper = v == "auto" ? CRD_STEPPER_AUTO : v == "staged" ? CRD_STEPPER_STAGED : v == "fused" ? CRD_STEPPER_FUSED : -2;
			} else if (!s.empty() && s[0] == '-') usage(argv[0], false);
			else if (o.ini.empty()) o.ini = s;
			else usage(argv[0], false);
		}
		if (o.model < 0 || o.surface < 0 || o.ini.empty() || o.precision == -2 || o.stepper == -2) usage(argv[0], false);
	}

	// Started by an MPI launcher the way the reference is (`mpirun -np N <exe> <ini>`, util/ShellScripts/run*.sh)?  This
	// program needs one process only: rank 0 drives N phi-slabs, one per GPU, and writes the N subdomain file sets the N
	// reference ranks would; the other ranks have nothing to do.
	const Launcher launcher = detect_launcher();
	if (launcher.size > 1 && launcher.rank != 0) return 0;

	crd_run_config cfg;
	char err[512];
	int rc = crd_config_load_ini(o.ini.c_str(), o.model, o.surface, &cfg, err, sizeof err);
	if (rc != CRD_OK) {
		std::cerr << "\nCRD_ERROR: cannot use " << o.ini << ": " << err << "\n\n";
		return 1;
	}
	if (o.gpus > 0) cfg.n_gpus = o.gpus;
	else if (launcher.size > 1 && cfg.n_gpus == 1) {
		cfg.n_gpus = launcher.size;
		if (o.devices <= 0) o.devices = std::max(1, std::min(launcher.size, crd_device_count()));
		if (!o.quiet)
			std::cout << "\n" << launcher.kind << " started " << launcher.size << " ranks: rank 0 drives " << launcher.size << " phi-slabs on " << o.devices
			          << " GPU(s), the other ranks exit\n";
	}
	if (o.dt > 0) cfg.dt = o.dt;
	if (o.stepper >= 0) cfg.stepper = o.stepper;
	if (o.precision >= 0) cfg.params.precision = o.precision;
	if (o.adaptive >= 0) cfg.adaptive = o.adaptive;
	if (o.ref_steady_state) cfg.steady_state_decimals = 8;  // numpy's print precision, util/GoldbeterModel/SolveGoldbeterODE.py:111
	time_t start_t = 0, end_t = 0;
	double total_t = 0, eta = 0;
	time(&start_t);

	crd_grid g;
	if ((rc = crd_grid_from_params(&cfg.params, &g)) != CRD_OK) return die("crd_grid_from_params", rc, nullptr);
	// Two decompositions.  FILES: phi-slabs (1 x gpus) unless the reference's own 2-D blocks are asked for -- `--decomp 2x2`, or
	// `--decomp mpi` = what MPI_Dims_create makes of the slab count (src/FHNmodel_torus.cpp:724-728: `mpirun -np 4` is 2 x 2): the file
	// sets, subdomain headers and (for the rand() rule) initial conditions of that process grid.  COMPUTE: phi-slabs, one per GPU (the
	// layout for one node, and the one the one-launch stepper, the error-controlled integrators and the deep halo need) -- also
	// under --decomp (round 4): the blocks' rows are cut out of the slabs' frames at output time; `--block-contexts` computes on
	// the 2-D blocks themselves instead (staged kernels, fixed step, text files), bit-equal to the whole domain's staged run.
	int fd0 = 1, fd1 = cfg.n_gpus;  // the files' process grid
	if (o.decomp_mpi) crd_dims_create(cfg.n_gpus, &fd0, &fd1);
	else if (o.d0 > 0) {
		fd0 = o.d0;
		fd1 = o.d1;
	}
	const int F = fd0 * fd1;
	const bool recut = fd0 > 1 && !o.block_contexts;  // files in a layout the contexts do not have
	const int d0 = recut ? 1 : fd0, d1 = recut ? cfg.n_gpus : fd1;  // the contexts' decomposition
	const int G = d0 * d1;
	if (fd0 > 1 && o.binary) {
		std::cerr << "\nCRD_ERROR: the .npy side-channel holds phi-slab frames: not available with theta-blocks (--decomp with more than one theta-block)\n\n";
		return 1;
	}
	if (d0 > 1 && cfg.adaptive) {
		std::cerr << "\nCRD_ERROR: theta-block contexts (--block-contexts) step with the fixed-step staged RK4 and write text files only\n\n";
		return 1;
	}
	const int ndev = o.devices > 0 ? o.devices : G;

	double s0 = 0, s1 = 0;  // banner only, and only printed for a constant beta (src/FHNmodel_torus.cpp:268-271)
	if (cfg.params.vary_beta == 0 && (rc = crd_steady_state_as_printed(cfg.params.model, cfg.params.beta, cfg.steady_state_decimals, &s0, &s1)) != CRD_OK)
		return die("crd_steady_state", rc, nullptr);

	// Output cadence (src/FHNmodel_torus.cpp:415-429): Nt outputs dTout apart; every interval is an integer number of
	// equal RK4 steps no longer than the requested / stable step.
	const int Nt = cfg.output_timestep;
	const double dTout = cfg.t_final / Nt;
	const double dt_cap = cfg.dt > 0 ? cfg.dt : cfg.dt_safety * crd_stable_dt(&cfg.params);
	const int64_t steps_per_output = (int64_t)std::ceil(dTout / dt_cap - 1e-12);
	const double dt = dTout / (double)steps_per_output;

	std::vector<crd_ctx *> ctx((size_t)G, nullptr);
	std::vector<crd_writer *> wr((size_t)F, nullptr);
	// recut: the frames of the F file blocks, cut out of the G slabs' frames (two sets, like the slabs' own)
	std::vector<std::vector<double>> fhost((size_t)(recut ? F : 0)), fhost_b((size_t)(recut ? F : 0));
	struct Rect {
		int64_t is, ie, js, je;
	};
	std::vector<Rect> frect((size_t)F), crect((size_t)G);
	// copy between the slabs' AoS frames (full width, rows cjs .. cje) and the blocks' (columns is .. ie of rows js .. je)
	auto recut_frames = [&](std::vector<std::vector<double>> &slabs, std::vector<std::vector<double>> &blocks, bool to_blocks) {
		for (int f = 0; f < F; f++) {
			const Rect &b = frect[(size_t)f];
			const int64_t nxl = b.ie - b.is + 1;
			for (int64_t j = b.js; j <= b.je; j++) {
				int c = 0;
				while (c + 1 < G && j > crect[(size_t)c].je) c++;
				double *slab_row = slabs[(size_t)c].data() + 2 * ((j - crect[(size_t)c].js) * g.nx + b.is), *block_row = blocks[(size_t)f].data() + 2 * (j - b.js) * nxl;
				if (to_blocks) std::memcpy(block_row, slab_row, (size_t)(2 * nxl) * sizeof(double));
				else std::memcpy(slab_row, block_row, (size_t)(2 * nxl) * sizeof(double));
			}
		}
	};
	// Two host copies of every slab: while the writer thread formats output k from one, the GPU integrates towards
	// output k+1 and downloads into the other (text output of a large grid costs more than the steps between outputs).
	std::vector<std::vector<double>> host((size_t)G), host_b((size_t)G);
	std::thread writer;
	int writer_rc = CRD_OK;
	// Binary side-channel: the owned rows of a field plane ARE the (nyl, nxl) frame of the .npy file, so a frame is one
	// device-to-host copy into page-locked memory (two sets: one being written to disk, one being filled) -- no layout change,
	// no formatting.  [slab][var][set]
	const int nvars_out = 1 + (cfg.include_all_vars == 1 ? 1 : 0);
	const size_t value_bytes = cfg.params.precision == CRD_PRECISION_F64 ? 8 : 4;
	std::vector<crd_npy_writer *> npy((size_t)(2 * G), nullptr);
	std::vector<void *> frame((size_t)(4 * G), nullptr);
	auto cleanup = [&]() {
		if (writer.joinable()) writer.join();
		for (auto *w : wr) crd_writer_close(w);
		for (auto *w : npy) crd_npy_writer_close(w);
		for (void *f : frame) crd_host_free(f);
		for (auto *c : ctx) crd_destroy(c);
	};
	for (int k = 0; k < G; k++) {
		if ((rc = crd_create_block(&cfg.params, k / d1, d0, k % d1, d1, k % ndev, &ctx[(size_t)k])) != CRD_OK) {
			die("crd_create", rc, nullptr);
			cleanup();
			return 1;
		}
		crd_set_stepper(ctx[(size_t)k], cfg.stepper);
	}
	if ((rc = crd_comm_attach_local(ctx.data(), G)) != CRD_OK) {
		die("crd_comm_attach_local", rc, ctx[0]);
		cleanup();
		return 1;
	}
	if (G > 1 && d0 == 1 && cfg.exchange_period) {
		// exchange period of the one-launch stepper: the ini's; without one the contexts keep what crd_create chose (10 steps where every
		// slab has 256 rows or more, else 8: include/crd.h, crd_set_exchange_period)
		for (int k = 0; k < G; k++)
			if ((rc = crd_set_exchange_period(ctx[(size_t)k], cfg.exchange_period)) != CRD_OK) {
				die("crd_set_exchange_period", rc, ctx[(size_t)k]);
				cleanup();
				return 1;
			}
	}

	for (int k = 0; k < G; k++) crd_get_block(ctx[(size_t)k], &crect[(size_t)k].is, &crect[(size_t)k].ie, &crect[(size_t)k].js, &crect[(size_t)k].je);
	for (int f = 0; f < F; f++) {
		if (recut) crd_block_extents(g.nx, g.ny, f / fd1, fd0, f % fd1, fd1, &frect[(size_t)f].is, &frect[(size_t)f].ie, &frect[(size_t)f].js, &frect[(size_t)f].je);
		else frect[(size_t)f] = crect[(size_t)f];
	}
	if (!o.quiet) banner(cfg, g, F, frect[0].ie - frect[0].is + 1, frect[0].je - frect[0].js + 1, s0, s1, dt, steps_per_output);

	// Initial conditions, subdomain files, first output row (src/FHNmodel_torus.cpp:285-354,376-410).  The initial conditions are
	// those of the FILES' process grid (the rand() rule draws block by block, as every reference rank does); recut: the blocks'
	// frames are then laid into the slabs' for the upload.
	for (int k = 0; k < G; k++) {
		host[(size_t)k].resize((size_t)(2 * (crect[(size_t)k].ie - crect[(size_t)k].is + 1) * (crect[(size_t)k].je - crect[(size_t)k].js + 1)));
		host_b[(size_t)k].resize(host[(size_t)k].size());
	}
	for (int f = 0; f < F; f++) {
		const Rect &b = frect[(size_t)f];
		std::vector<double> &ic = recut ? fhost[(size_t)f] : host[(size_t)f];
		if (recut) {
			fhost[(size_t)f].resize((size_t)(2 * (b.ie - b.is + 1) * (b.je - b.js + 1)));
			fhost_b[(size_t)f].resize(fhost[(size_t)f].size());
		}
		if ((rc = crd_initial_conditions_block(&cfg, b.is, b.ie, b.js, b.je, ic.data())) != CRD_OK) {
			die("crd_initial_conditions", rc, nullptr);
			cleanup();
			return 1;
		}
		if ((rc = crd_writer_open_block(&cfg, o.outdir.c_str(), f, f / fd1, fd0, f % fd1, fd1, &wr[(size_t)f])) != CRD_OK ||
		    (!o.binary_only && (rc = crd_writer_write_row(wr[(size_t)f], ic.data())) != CRD_OK)) {
			die("crd_writer", rc, nullptr);
			cleanup();
			return 1;
		}
	}
	if (recut) recut_frames(host, fhost, false);
	for (int k = 0; k < G; k++) {
		const int64_t js = crect[(size_t)k].js, je = crect[(size_t)k].je;
		if ((rc = crd_state_upload(ctx[(size_t)k], host[(size_t)k].data(), 1)) != CRD_OK) {
			die("crd_state_upload", rc, ctx[(size_t)k]);
			cleanup();
			return 1;
		}
		if (o.binary) {
			const size_t frame_bytes = (size_t)(g.nx * (je - js + 1)) * value_bytes;
			for (int v = 0; v < nvars_out && rc == CRD_OK; v++) {
				rc = crd_npy_writer_open(&cfg, o.outdir.c_str(), k, G, v, (int)value_bytes, &npy[(size_t)(2 * k + v)]);
				for (int set = 0; set < 2 && rc == CRD_OK; set++)
					if (!(frame[(size_t)(4 * k + 2 * v + set)] = crd_host_alloc(frame_bytes))) rc = CRD_ENOMEM;
				if (rc == CRD_OK) rc = crd_state_download_rows(ctx[(size_t)k], v, 0, je - js + 1, frame[(size_t)(4 * k + 2 * v)]);
				if (rc == CRD_OK) rc = crd_npy_writer_append(npy[(size_t)(2 * k + v)], frame[(size_t)(4 * k + 2 * v)]);  // the initial state
			}
			if (rc != CRD_OK) {
				die("crd_npy_writer", rc, ctx[(size_t)k]);
				cleanup();
				return 1;
			}
		}
	}

	// The one-launch stepper measures its launch plan on a context's first full-size step (~0.7 s at 8192^2): done here, ahead of the
	// first output interval, so that the rate line below is the stepping's and nothing else's.  (The error-controlled integrators
	// measure theirs inside their first attempt.)
	if (!cfg.adaptive) {
		const auto plan_t0 = std::chrono::steady_clock::now();
		for (int k = 0; k < G; k++)
			if ((rc = crd_plan_launches(ctx[(size_t)k])) != CRD_OK) {
				die("crd_plan_launches", rc, ctx[(size_t)k]);
				cleanup();
				return 1;
			}
		const double plan_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - plan_t0).count();
		crd_launch_plan lp;
		if (!o.quiet && crd_get_launch_plan(ctx[0], &lp) == CRD_OK && lp.tuned)
			std::printf("   launch plan (measured in %.2f s): chunk mode %d, XCD mapping %d, %d column(s) per lane, %s stores, %d step(s) per launch\n", plan_s, lp.one_round,
			            lp.xcd_mapping, lp.columns_per_lane, lp.nontemporal_stores ? "non-temporal" : "plain", lp.steps_per_launch);
	}

	int status = 0;
	double adaptive_h = 0.0;
	long long adaptive_steps = 0, adaptive_rejected = 0;
	// Rate summary ("keep banner; add steps/s, point-steps/s, GB/s"): wall time spent inside the stepping calls
	// (they block on the download that follows, so the device work of an interval is inside its bracket) and the steps taken.
	double stepping_s = 0.0;
	long long steps_taken = 0;
	for (int iout = 0; iout < Nt; iout++) {
		const double t = iout * dTout;
		char range_name[64];
		std::snprintf(range_name, sizeof range_name, "crd_run output interval %d/%d", iout + 1, Nt);
		crd_trace_range_push(range_name);
		const auto step_t0 = std::chrono::steady_clock::now();
		auto &buf = (iout & 1) ? host_b : host;  // the other set may still be in the writer's hands
		if (cfg.adaptive) {
			// one ARKode(...) call per output interval, src/FHNmodel_torus.cpp:423; the controller's step carries over
			crd_adaptive_options ao;
			crd_adaptive_defaults(&ao);
			ao.rtol = cfg.rtol;
			ao.atol = cfg.atol;
			ao.method = cfg.adaptive == 2 ? CRD_ADAPT_RK43 : CRD_ADAPT_ARKODE;
			ao.h0 = cfg.adaptive == 2 ? adaptive_h : 0.0;  // (the ARKode-style controller keeps its own memory in the contexts; first step: arkHin)
			ao.dense_output = 1;  // ARK_NORMAL: output times do not shorten steps, the row written is the interpolant at tout
			crd_adaptive_stats as;
			rc = crd_group_integrate_adaptive(ctx.data(), G, t, (iout + 1 == Nt) ? cfg.t_final : (iout + 1) * dTout, &ao, &as);
			adaptive_h = as.h_next;
			adaptive_steps += as.accepted;
			adaptive_rejected += as.rejected;
			steps_taken += as.accepted + as.rejected;
		} else {
			rc = crd_group_step_rk4(ctx.data(), G, t, dt, steps_per_output);
			steps_taken += steps_per_output;
		}
		for (int k = 0; k < G && rc == CRD_OK; k++) rc = crd_synchronize(ctx[(size_t)k]);
		stepping_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - step_t0).count();
		const int set = (iout + 1) & 1;  // the other set of frame buffers may still be in the writer's hands
		for (int k = 0; k < G && rc == CRD_OK; k++) {
			if (!o.binary_only) rc = crd_state_download(ctx[(size_t)k], buf[(size_t)k].data(), 1);
			int64_t js, je;
			crd_get_slab(ctx[(size_t)k], &js, &je);
			for (int v = 0; o.binary && v < nvars_out && rc == CRD_OK; v++)
				rc = crd_state_download_rows(ctx[(size_t)k], v, 0, je - js + 1, frame[(size_t)(4 * k + 2 * v + set)]);
		}
		// the reference stops at the first failing ARKode call on ANY rank (src/FHNmodel_torus.cpp:424-435): look at every slab
		bool blown = false;  // sticky: a NaN / Inf on ANY slab stops the run, whatever the later slabs hold
		for (int k = 0; k < G && rc == CRD_OK; k++) {
			double pk = 0;
			rc = crd_state_max_abs(ctx[(size_t)k], &pk);
			blown = blown || !std::isfinite(pk);
		}
		if (rc != CRD_OK || blown) {
			if (rc != CRD_OK) die("crd_group_step_rk4", rc, ctx[0]);
			std::cerr << "Solver failure, stopping integration\n";  // src/FHNmodel_torus.cpp:433
			status = 1;
			crd_trace_range_pop();
			break;
		}
		if (writer.joinable()) writer.join();
		if (writer_rc != CRD_OK) {
			die("crd_writer_write_row", writer_rc, nullptr);
			status = 1;
			crd_trace_range_pop();
			break;
		}
		auto &fbuf = (iout & 1) ? fhost_b : fhost;
		writer = std::thread([&wr, &buf, &fbuf, &writer_rc, &npy, &frame, &o, &recut_frames, G, F, recut, set, nvars_out]() {
			for (int k = 0; k < G && writer_rc == CRD_OK; k++)
				for (int v = 0; o.binary && v < nvars_out && writer_rc == CRD_OK; v++)
					writer_rc = crd_npy_writer_append(npy[(size_t)(2 * k + v)], frame[(size_t)(4 * k + 2 * v + set)]);
			if (recut && !o.binary_only) recut_frames(buf, fbuf, true);  // the file blocks' frames out of the slabs' (on the writer's time)
			for (int f = 0; f < F && !o.binary_only && writer_rc == CRD_OK; f++)
				writer_rc = crd_writer_write_row(wr[(size_t)f], (recut ? fbuf : buf)[(size_t)f].data());
		});

		// progress line, src/FHNmodel_torus.cpp:457-477
		time(&end_t);
		total_t += difftime(end_t, start_t);
		start_t = end_t;
		eta = (Nt - (iout + 1)) * (total_t / (iout + 1));
		if (!o.quiet) {
			if (iout > 0) std::printf("\r");
			std::printf("   %3d %% | %3d min %2d sec elapsed | %3d min %2d sec remaining", 100 * (iout + 1) / Nt, (int)(total_t / 60), ((int)total_t % 60),
			            (int)(eta / 60), ((int)eta % 60));
			std::fflush(stdout);
		}
		crd_trace_range_pop();
	}
	if (writer.joinable()) writer.join();
	if (writer_rc != CRD_OK && status == 0) status = die("crd_writer_write_row", writer_rc, nullptr);
	if (!o.quiet && cfg.adaptive) std::cout << "\n   steps = " << adaptive_steps << " (+" << adaptive_rejected << " rejected)";
	if (!o.quiet && steps_taken > 0 && stepping_s > 0.0) {
		// compulsory-byte model of a step: both fields read once and written once (the one-launch stepper's traffic; the four
		// stage kernels move 8 x that)
		crd_launch_plan lp{};
		const bool pairs = !cfg.adaptive && crd_get_launch_plan(ctx[0], &lp) == CRD_OK && lp.tuned && lp.steps_per_launch == 2;
		const double points = (double)g.nx * (double)g.ny, bytes_per_point_step = 4.0 * (double)value_bytes / (pairs ? 2.0 : 1.0);  // (two steps per launch: the state crosses memory once per two)
		char line[256];
		std::snprintf(line, sizeof line, "\n   rate: %lld steps in %.6f s of stepping = %.1f steps/s, %.4g grid-point-steps/s, %.1f GB/s (%g B per point-step)",
		              steps_taken, stepping_s, (double)steps_taken / stepping_s, points * (double)steps_taken / stepping_s,
		              points * (double)steps_taken * bytes_per_point_step / stepping_s / 1e9, bytes_per_point_step);
		std::cout << line;
	}
	if (!o.quiet) std::cout << "\n   ----------------------\n";
	cleanup();
	return status;
}
