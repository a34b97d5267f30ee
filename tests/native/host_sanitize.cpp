// Exercises the host-only entry points of libcrd (crd_host.cpp, crd_io.cpp: no HIP involved) in a build with
// AddressSanitizer and UndefinedBehaviorSanitizer.  Built and run by tests/test_host_sanitizers.py with g++.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "crd.h"

#define CHECK(cond)                                                              \
	do {                                                                         \
		if (!(cond)) {                                                           \
			std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); \
			return 1;                                                            \
		}                                                                        \
	} while (0)

static void write_file(const std::string &path, const std::string &text)
{
	FILE *f = std::fopen(path.c_str(), "w");
	std::fputs(text.c_str(), f);
	std::fclose(f);
}

int main(int argc, char **argv)
{
	CHECK(argc == 2);
	const std::string dir = argv[1];
	CHECK(crd_abi_version() == CRD_ABI_VERSION);
	for (int st = 1; st >= -9; st--) CHECK(crd_status_string(st) != nullptr);

	// ---- ini files: a complete one, odd spacing / comments, missing key, malformed number, missing file, tiny err buffer
	const std::string good = dir + "/good.ini";
	write_file(good,
	           "; comment\n[Parameters]\ndiffusion = 0.12\nbeta=1.25\n surfaceWidth =20\nsurfaceLength = 80\nwaveLength = 0.1\nwaveWidth = 0.5\n"
	           "waveInside = 0\noutputTimestep = 3\ntBoundary = 0.4\ntFinal = 1.2\nthetaMesh = 16\nphiMesh = 40\nbetaMin = 0.7\nbetaMax = 1.7\n\n"
	           "[System]\nincludeAllVars = 1\nvaryBeta = 0\n[Solver]\ndt = 0.02\ngpus = 2\nstepper = 2\nadaptive = 1\nrtol=1e-6\n");
	crd_run_config cfg;
	char err[256];
	if (crd_config_load_ini(good.c_str(), CRD_MODEL_FHN, CRD_SURFACE_TORUS, &cfg, err, sizeof err) != CRD_OK) {
		std::fprintf(stderr, "good.ini refused: %s\n", err);
		return 1;
	}
	CHECK(cfg.params.nx == 16 && cfg.params.ny == 40 && cfg.n_gpus == 2 && cfg.adaptive == 1 && cfg.include_all_vars == 1);
	CHECK(crd_config_load_ini(good.c_str(), CRD_MODEL_FHN, CRD_SURFACE_TORUS, &cfg, nullptr, 0) == CRD_OK);
	write_file(dir + "/missing.ini", "[Parameters]\ndiffusion = 0.12\n[System]\nvaryBeta = 0\n");
	char tiny[4];
	CHECK(crd_config_load_ini((dir + "/missing.ini").c_str(), CRD_MODEL_FHN, CRD_SURFACE_TORUS, &cfg, tiny, sizeof tiny) == CRD_EPARSE);
	CHECK(std::strlen(tiny) < sizeof tiny);
	write_file(dir + "/bad.ini", "[Parameters]\ndiffusion = abc\nbeta = \nthetaMesh = 1e400\n[System\nvaryBeta 0\n=\n[]\n");
	CHECK(crd_config_load_ini((dir + "/bad.ini").c_str(), CRD_MODEL_GOLDBETER, CRD_SURFACE_FLAT, &cfg, err, sizeof err) != CRD_OK);
	CHECK(crd_config_load_ini((dir + "/nope.ini").c_str(), CRD_MODEL_FHN, CRD_SURFACE_TORUS, &cfg, err, sizeof err) != CRD_OK);
	write_file(dir + "/empty.ini", "");
	CHECK(crd_config_load_ini((dir + "/empty.ini").c_str(), CRD_MODEL_FHN, CRD_SURFACE_FLAT, &cfg, err, sizeof err) != CRD_OK);
	CHECK(crd_config_load_ini(nullptr, CRD_MODEL_FHN, CRD_SURFACE_FLAT, &cfg, err, sizeof err) != CRD_OK);

	// ---- ini reader under mutation: 3000 seeded random edits of the good file (bytes replaced, lines cut, duplicated, truncated);
	//      any status is fine, a crash or a sanitizer report is not; whatever it accepts must describe a sane run
	{
		std::string base;
		{
			FILE *f = std::fopen(good.c_str(), "r");
			CHECK(f);
			char buf[4096];
			const size_t k = std::fread(buf, 1, sizeof buf, f);
			std::fclose(f);
			base.assign(buf, k);
		}
		unsigned long long s = 0x9E3779B97F4A7C15ull;
		auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
		const char alphabet[] = "=[]\n\t ;#.-+eE0123456789abcxyzParametersSystemSolver\0\xff";
		int accepted = 0;
		for (int it = 0; it < 3000; it++) {
			std::string t = base;
			const int edits = 1 + (int)(rnd() % 4);
			for (int e = 0; e < edits && !t.empty(); e++) {
				const size_t at = rnd() % t.size();
				switch (rnd() % 5) {
				case 0: t[at] = alphabet[rnd() % (sizeof alphabet - 1)]; break;
				case 1: t.erase(at, 1 + rnd() % 12); break;
				case 2: t.insert(at, t.substr(rnd() % t.size(), rnd() % 40)); break;
				case 3: t.resize(at); break;
				default: t.insert(at, 1, alphabet[rnd() % (sizeof alphabet - 1)]); break;
				}
			}
			write_file(dir + "/mut.ini", t);
			crd_run_config c;
			char e2[64];
			const int model = (int)(rnd() % 2), surface = (int)(rnd() % 2);
			if (crd_config_load_ini((dir + "/mut.ini").c_str(), model, surface, &c, e2, sizeof e2) == CRD_OK) {
				accepted++;
				crd_grid gg;
				CHECK(crd_grid_from_params(&c.params, &gg) == CRD_OK && gg.nx >= 2 && gg.ny >= 2 && c.output_timestep >= 1 && c.t_final > 0.0);
			}
			CHECK(std::strlen(e2) < sizeof e2);
		}
		CHECK(accepted > 0 && accepted < 3000);
	}

	// ---- geometry, slabs, plans
	CHECK(crd_config_load_ini(good.c_str(), CRD_MODEL_FHN, CRD_SURFACE_TORUS, &cfg, err, sizeof err) == CRD_OK);
	crd_grid g;
	CHECK(crd_grid_from_params(&cfg.params, &g) == CRD_OK && g.nx == 16 && g.ny == 40);
	crd_params derived = cfg.params;
	derived.ny = 0;
	derived.nx = 100;
	derived.surface_length = 100.0;
	CHECK(crd_grid_from_params(&derived, &g) == CRD_OK && g.ny == 499);  // (long)(100 * (R / r)) truncates, SURVEY 8c G8
	crd_params bad = cfg.params;
	bad.nx = 0;
	CHECK(crd_grid_from_params(&bad, &g) != CRD_OK);
	CHECK(crd_grid_from_params(nullptr, &g) != CRD_OK);
	for (int n = 1; n <= 7; n++) {
		int64_t covered = 0;
		for (int k = 0; k < n; k++) {
			int64_t js, je;
			CHECK(crd_slab_extents(40, k, n, &js, &je) == CRD_OK && js == covered);
			covered = je + 1;
			crd_halo_op ops[4];
			CHECK(crd_halo_plan(k, n, je - js + 1, 1, ops) == CRD_OK);
			int sends = 0;
			for (const crd_halo_op &op : ops) sends += op.is_send;
			CHECK(sends == 2);
		}
		CHECK(covered == 40);
	}
	int64_t js, je;
	CHECK(crd_slab_extents(40, 3, 3, &js, &je) != CRD_OK && crd_slab_extents(40, -1, 3, &js, &je) != CRD_OK && crd_slab_extents(2, 0, 3, &js, &je) != CRD_OK);
	crd_halo_op ops[4];
	CHECK(crd_halo_plan(0, 2, 4, 5, ops) != CRD_OK);  // deeper than the slab

	// ---- steady states, stable step, initial conditions of all four programs
	double s0, s1;
	CHECK(crd_steady_state(CRD_MODEL_FHN, 1.25, &s0, &s1) == CRD_OK && std::fabs(s0 + 1.25) < 1e-15);
	CHECK(crd_steady_state(CRD_MODEL_GOLDBETER, 0.4, &s0, &s1) == CRD_OK && s0 > 0 && s1 > 0);
	CHECK(crd_steady_state(7, 0.4, &s0, &s1) != CRD_OK);
	CHECK(crd_stable_dt(&cfg.params) > 0.0);
	for (int model : {CRD_MODEL_FHN, CRD_MODEL_GOLDBETER})
		for (int surface : {CRD_SURFACE_TORUS, CRD_SURFACE_FLAT})
			for (int vary : {0, 1})
				for (int ic : {0, 1, 2}) {
					crd_run_config c = cfg;
					c.params.model = model;
					c.params.surface = surface;
					c.params.vary_beta = vary;
					c.params.beta = model == CRD_MODEL_FHN ? 1.25 : 0.4;
					c.params.ny = 24;
					c.params.surface_length = 20.0;
					c.ic_type = ic;
					std::vector<double> y(2 * 16 * 7);
					CHECK(crd_initial_conditions(&c, 5, 11, y.data()) == CRD_OK);
					for (double v : y) CHECK(std::isfinite(v));
					CHECK(crd_initial_conditions(&c, 11, 5, y.data()) != CRD_OK);
					CHECK(crd_initial_conditions(&c, 0, 24, y.data()) != CRD_OK);
				}

	// ---- writer: two slabs, three rows, both variables; then a refused directory
	for (int k = 0; k < 2; k++) {
		crd_writer *w = nullptr;
		CHECK(crd_writer_open(&cfg, dir.c_str(), k, 2, &w) == CRD_OK && w);
		CHECK(crd_slab_extents(40, k, 2, &js, &je) == CRD_OK);
		std::vector<double> y((size_t)2 * 16 * (size_t)(je - js + 1));
		for (int t = 0; t < 3; t++) {
			for (size_t q = 0; q < y.size(); q++) y[q] = (q % 7 == 0 ? -1.0 : 1.0) * std::ldexp(1.0 + 1e-3 * (double)q, (int)(q % 200) - 100 + t);
			y[1] = 0.0;
			y[2] = -0.0;
			y[3] = 5e-324;
			y[4] = 1.7976931348623157e308;
			CHECK(crd_writer_write_row(w, y.data()) == CRD_OK);
		}
		CHECK(crd_writer_close(w) == CRD_OK);
	}
	{  // a slab large enough (> 65536 values per row) that the writer formats on several threads
		crd_run_config big = cfg;
		big.params.nx = 512;
		big.params.ny = 600;
		big.params.model = CRD_MODEL_GOLDBETER;
		crd_writer *wb = nullptr;
		CHECK(crd_writer_open(&big, dir.c_str(), 0, 1, &wb) == CRD_OK && wb);
		std::vector<double> y((size_t)2 * 512 * 600);
		for (size_t q = 0; q < y.size(); q++) y[q] = std::sin(0.001 * (double)q) * std::ldexp(1.0, (int)(q % 61) - 30);
		CHECK(crd_writer_write_row(wb, y.data()) == CRD_OK && crd_writer_write_row(wb, y.data()) == CRD_OK);
		CHECK(crd_writer_close(wb) == CRD_OK);
		// read the first and the last value of the first row back
		FILE *f = std::fopen((dir + "/GoldbeterModel_torus_Z.000.txt").c_str(), "r");
		CHECK(f);
		double first = 0.0, v = 0.0;
		CHECK(std::fscanf(f, "%lf", &first) == 1 && first == y[0]);
		for (size_t q = 1; q < (size_t)512 * 600; q++) CHECK(std::fscanf(f, "%lf", &v) == 1);
		CHECK(v == y[2 * ((size_t)512 * 600 - 1)]);
		std::fclose(f);
	}
	{  // every value at the widest text form, 25 characters (" -1.5000000000000000e-200"): a whole multi-thread row of them, then
	   // +-1e+-200, subnormals and non-finite values mixed; each row must read back exactly
		crd_run_config wide = cfg;
		wide.params.nx = 512;
		wide.params.ny = 512;
		wide.params.model = CRD_MODEL_FHN;
		wide.params.surface = CRD_SURFACE_FLAT;
		wide.params.surface_length = wide.params.surface_width = 20.0;
		wide.include_all_vars = 1;
		crd_writer *ww = nullptr;
		CHECK(crd_writer_open(&wide, dir.c_str(), 0, 1, &ww) == CRD_OK && ww);
		std::vector<double> y((size_t)2 * 512 * 512, -1.5e-200);
		CHECK(crd_writer_write_row(ww, y.data()) == CRD_OK);
		const double odd[8] = {-1e-200, -1e200, 1e-200, 1e200, -4.9406564584124654e-324, -1.7976931348623157e308, -INFINITY, NAN};
		for (size_t q = 0; q < y.size(); q++) y[q] = odd[q % 8];
		CHECK(crd_writer_write_row(ww, y.data()) == CRD_OK);
		CHECK(crd_writer_close(ww) == CRD_OK);
		FILE *f = std::fopen((dir + "/FHNmodel_flat_u.000.txt").c_str(), "r");
		CHECK(f);
		double v = 0.0;
		for (size_t q = 0; q < (size_t)512 * 512; q++) CHECK(std::fscanf(f, "%lf", &v) == 1 && v == -1.5e-200);
		for (size_t q = 0; q < (size_t)512 * 512; q++) {
			CHECK(std::fscanf(f, "%lf", &v) == 1);
			const double want = odd[(2 * q) % 8];
			CHECK(std::isnan(want) ? std::isnan(v) : v == want);
		}
		std::fclose(f);
	}
	{  // binary side-channel: header, three frames, header rewritten at close
		crd_npy_writer *nw = nullptr;
		CHECK(crd_npy_writer_open(&cfg, dir.c_str(), 1, 2, 1, 8, &nw) == CRD_OK && nw);
		CHECK(crd_slab_extents(40, 1, 2, &js, &je) == CRD_OK);
		std::vector<double> fr((size_t)16 * (size_t)(je - js + 1), 1.5);
		for (int t = 0; t < 3; t++) CHECK(crd_npy_writer_append(nw, fr.data()) == CRD_OK);
		CHECK(crd_npy_writer_append(nw, nullptr) != CRD_OK);
		CHECK(crd_npy_writer_close(nw) == CRD_OK);
		FILE *f = std::fopen((dir + "/FHNmodel_torus_v.001.npy").c_str(), "rb");
		CHECK(f);
		char head[128];
		CHECK(std::fread(head, 1, sizeof head, f) == sizeof head && std::memcmp(head, "\x93NUMPY", 6) == 0 && head[127] == '\n');
		CHECK(std::string(head + 10, 117).find("'shape': (3, 20, 16)") != std::string::npos);
		std::fseek(f, 0, SEEK_END);
		CHECK(std::ftell(f) == 128 + 3 * 20 * 16 * 8);
		std::fclose(f);
		CHECK(crd_npy_writer_open(&cfg, dir.c_str(), 0, 1, 0, 5, &nw) != CRD_OK && crd_npy_writer_close(nullptr) == CRD_OK);
	}
	crd_writer *w = nullptr;
	CHECK(crd_writer_open(&cfg, (dir + "/no/such/dir").c_str(), 0, 1, &w) != CRD_OK && w == nullptr);
	CHECK(crd_writer_open(&cfg, dir.c_str(), 2, 2, &w) != CRD_OK);
	CHECK(crd_writer_close(nullptr) != CRD_OK || true);
	std::puts("host sanitize run ok");
	return 0;
}
