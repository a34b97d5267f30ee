// crd_io.cpp -- the two file formats libcrd shares with the reference: the .ini parameter file it reads and the
// per-subdomain text files it writes (consumed unmodified by util/*/plot_*.py and MapOutputToTorus.py).
#include <algorithm>
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <thread>

#include <fcntl.h>
#include <unistd.h>

#include "crd_internal.h"

namespace {

std::string trim(const std::string &s)
{
	size_t a = 0, b = s.size();
	while (a < b && std::isspace((unsigned char)s[a])) a++;
	while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
	return s.substr(a, b - a);
}

// Minimal INI reader with the rules of boost::property_tree::ini_parser that the shipped files rely on
// (data/FHNmodelArgs.ini:1-20): '#' or ';' starts a comment line, "[Section]", "key = value" with both sides
// trimmed (the shipped values carry trailing tabs), duplicate keys and lines without '=' are errors.
struct Ini {
	std::map<std::string, std::string> kv;  // "Section.key" -> value

	bool load(const char *path, std::string *err)
	{
		std::ifstream in(path);
		if (!in) {
			*err = std::string("cannot open ") + path;
			return false;
		}
		std::string line, section;
		int lineno = 0;
		while (std::getline(in, line)) {
			lineno++;
			line = trim(line);
			if (line.empty() || line[0] == '#' || line[0] == ';') continue;
			if (line[0] == '[') {
				if (line.back() != ']') {
					*err = "unmatched '[' at line " + std::to_string(lineno);
					return false;
				}
				section = trim(line.substr(1, line.size() - 2));
				continue;
			}
			const size_t eq = line.find('=');
			if (eq == std::string::npos) {
				*err = "'=' character not found at line " + std::to_string(lineno);
				return false;
			}
			const std::string key = trim(line.substr(0, eq));
			if (key.empty()) {
				*err = "key expected at line " + std::to_string(lineno);
				return false;
			}
			const std::string full = section.empty() ? key : section + "." + key;
			if (kv.count(full)) {
				*err = "duplicate key name '" + full + "' at line " + std::to_string(lineno);
				return false;
			}
			kv[full] = trim(line.substr(eq + 1));
		}
		return true;
	}

	bool has(const std::string &k) const { return kv.count(k) != 0; }

	// pt.get<T>(): the whole value must convert (boost's stream_translator rejects trailing characters).
	bool get_double(const std::string &k, double *out, std::string *err) const
	{
		auto it = kv.find(k);
		if (it == kv.end()) {
			*err = "No such node (" + k + ")";
			return false;
		}
		std::istringstream iss(it->second);
		iss >> *out;
		if (iss.fail()) {
			*err = "conversion of data to type \"double\" failed (" + k + " = " + it->second + ")";
			return false;
		}
		if (!iss.eof()) iss >> std::ws;
		if (!iss.eof()) {
			*err = "conversion of data to type \"double\" failed (" + k + " = " + it->second + ")";
			return false;
		}
		return true;
	}

	bool get_int(const std::string &k, long long *out, std::string *err) const
	{
		auto it = kv.find(k);
		if (it == kv.end()) {
			*err = "No such node (" + k + ")";
			return false;
		}
		std::istringstream iss(it->second);
		iss >> *out;
		if (!iss.fail() && !iss.eof()) iss >> std::ws;
		if (iss.fail() || !iss.eof()) {
			*err = "conversion of data to type \"int\" failed (" + k + " = " + it->second + ")";
			return false;
		}
		return true;
	}
};

void set_err(char *err, size_t len, const std::string &msg)
{
	if (err && len) {
		std::snprintf(err, len, "%s", msg.c_str());
	}
}

}  // namespace

extern "C" int crd_config_load_ini(const char *path, int model, int surface, crd_run_config *cfg, char *err, size_t err_len)
{
	if (!path || !cfg) return CRD_EINVAL;
	if ((model != CRD_MODEL_FHN && model != CRD_MODEL_GOLDBETER) || (surface != CRD_SURFACE_TORUS && surface != CRD_SURFACE_FLAT)) {
		set_err(err, err_len, "bad model / surface");
		return CRD_EINVAL;
	}
	Ini ini;
	std::string why;
	if (!ini.load(path, &why)) {
		set_err(err, err_len, why);
		return why.rfind("cannot open", 0) == 0 ? CRD_EIO : CRD_EPARSE;
	}
	std::memset(cfg, 0, sizeof(*cfg));
	crd_params &p = cfg->params;
	p.model = model;
	p.surface = surface;
	p.precision = CRD_PRECISION_F64;
	cfg->dt_safety = 0.8;
	cfg->n_gpus = 1;
	cfg->stepper = CRD_STEPPER_AUTO;
	cfg->adaptive = 0;
	cfg->rtol = 1.e-5;   // src/FHNmodel_torus.cpp:197-198
	cfg->atol = 1.e-10;

	bool ok = true;
	auto D = [&](const char *k, double *dst) { ok = ok && ini.get_double(std::string("Parameters.") + k, dst, &why); };
	long long iv = 0;
	auto I = [&](const std::string &k, int32_t *dst) {
		if (ok && (ok = ini.get_int(k, &iv, &why))) *dst = (int32_t)iv;
	};

	// Keys every program reads: src/FHNmodel_torus.cpp:160-169 and siblings.
	D("diffusion", &p.diffusion);
	D("beta", &p.beta);
	D("surfaceLength", &p.surface_length);
	D("surfaceWidth", &p.surface_width);
	D("waveLength", &cfg->wave_length);
	D("waveWidth", &cfg->wave_width);
	I("Parameters.outputTimestep", &cfg->output_timestep);
	D("tBoundary", &p.t_boundary);
	D("tFinal", &cfg->t_final);
	if (surface == CRD_SURFACE_TORUS) I("Parameters.waveInside", &cfg->wave_inside);  // torus programs only (:166)

	// Mesh key: FHN reads thetaMesh (src/FHNmodel_torus.cpp:170), Goldbeter xMesh (src/GoldbeterModel_torus.cpp:184);
	// the shipped data/FHNmodelArgs.ini carries xMesh, data/temp.ini thetaMesh, so either is accepted.
	if (ok) {
		const char *own = (model == CRD_MODEL_FHN) ? "Parameters.thetaMesh" : "Parameters.xMesh";
		const char *other = (model == CRD_MODEL_FHN) ? "Parameters.xMesh" : "Parameters.thetaMesh";
		const char *use = ini.has(own) ? own : (ini.has(other) ? other : own);
		if ((ok = ini.get_int(use, &iv, &why))) p.nx = iv;
	}
	// phiMesh / yMesh: extension; absent = derive ny like the reference.
	if (ok) {
		for (const char *k : {"Parameters.phiMesh", "Parameters.yMesh"}) {
			if (ini.has(k)) {
				if ((ok = ini.get_int(k, &iv, &why))) p.ny = iv;
				break;
			}
		}
	}

	I("System.includeAllVars", &cfg->include_all_vars);
	I("System.varyBeta", &p.vary_beta);

	// betaMin / betaMax: read by both FHN programs (:171-172) and by Goldbeter flat (src/GoldbeterModel_flat.cpp:179-180);
	// Goldbeter torus never reads them, its BETA_MIN / BETA_MAX stay 0 (src/GoldbeterModel_torus.cpp:101-102).
	// Lenient superset: mandatory only when varyBeta = 1 actually uses them.
	const bool reads_beta_range = !(model == CRD_MODEL_GOLDBETER && surface == CRD_SURFACE_TORUS);
	if (ok && reads_beta_range) {
		const bool needed = (p.vary_beta != 0);
		if (needed || ini.has("Parameters.betaMin")) D("betaMin", &p.beta_min);
		if (needed || ini.has("Parameters.betaMax")) D("betaMax", &p.beta_max);
	}
	if (model == CRD_MODEL_GOLDBETER) {
		I("System.justDiffusion", &p.just_diffusion);
		if (surface == CRD_SURFACE_FLAT) I("System.icType", &cfg->ic_type);  // src/GoldbeterModel_flat.cpp:184 only
	}

	// [Solver] extension section.
	if (ok && ini.has("Solver.dt")) ok = ini.get_double("Solver.dt", &cfg->dt, &why);
	if (ok && ini.has("Solver.dtSafety")) ok = ini.get_double("Solver.dtSafety", &cfg->dt_safety, &why);
	if (ok && ini.has("Solver.gpus")) I("Solver.gpus", &cfg->n_gpus);
	if (ok && ini.has("Solver.stepper")) I("Solver.stepper", &cfg->stepper);
	if (ok && ini.has("Solver.adaptive")) I("Solver.adaptive", &cfg->adaptive);
	if (ok && ini.has("Solver.steadyStateDigits")) I("Solver.steadyStateDigits", &cfg->steady_state_decimals);
	if (ok && ini.has("Solver.exchangePeriod")) I("Solver.exchangePeriod", &cfg->exchange_period);
	if (ok && ini.has("Solver.rtol")) ok = ini.get_double("Solver.rtol", &cfg->rtol, &why);
	if (ok && ini.has("Solver.atol")) ok = ini.get_double("Solver.atol", &cfg->atol, &why);
	if (ok && ini.has("Solver.precision")) {
		int32_t bits = 64;
		I("Solver.precision", &bits);
		if (ok && bits != 64 && bits != 32) {
			ok = false;
			why = "Solver.precision must be 64 or 32";
		}
		p.precision = (bits == 32) ? CRD_PRECISION_F32 : CRD_PRECISION_F64;
	}

	if (!ok) {
		set_err(err, err_len, why);
		return CRD_EPARSE;
	}
	if (!crd::validate_params(p, &why)) {
		set_err(err, err_len, why);
		return CRD_EINVAL;
	}
	if (cfg->output_timestep < 1 || !(cfg->t_final > 0.0) || cfg->n_gpus < 1 || !(cfg->dt >= 0.0) || !(cfg->dt_safety > 0.0) || !(cfg->rtol >= 0.0) ||
	    !(cfg->atol >= 0.0) || !(cfg->rtol + cfg->atol > 0.0)) {
		set_err(err, err_len, "outputTimestep, tFinal, gpus, dt, dtSafety, rtol or atol out of range");
		return CRD_EINVAL;
	}
	if (cfg->adaptive < 0 || cfg->adaptive > 2) {
		set_err(err, err_len, "Solver.adaptive must be 0 (fixed step), 1 (ARKode-style) or 2 (RK4(3))");
		return CRD_EINVAL;
	}
	if (cfg->exchange_period != 0 && (cfg->exchange_period < 3 || cfg->exchange_period > 16)) {
		set_err(err, err_len, "Solver.exchangePeriod must be 0 (automatic) or 3 .. 16");
		return CRD_EINVAL;
	}
	if (cfg->steady_state_decimals < 0 || cfg->steady_state_decimals > 17) {
		set_err(err, err_len, "Solver.steadyStateDigits must be 0 .. 17");
		return CRD_EINVAL;
	}
	set_err(err, err_len, "");
	return CRD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Text writer.  Header and rows as src/FHNmodel_torus.cpp:376-410,438-455: the subdomain file holds
// "nx  ny  is  ie  js  je xmin xmax tfinal"; each data row is nyl*nxl values " %.16e" (j outer, i inner) + "\n";
// the second-variable file is always created but only filled when includeAllVars == 1.
// ---------------------------------------------------------------------------------------------------------------
struct crd_writer {
	int f0 = -1;  // file descriptors, written with pwrite at the tracked offsets (no stdio copy)
	int f1 = -1;
	int64_t off0 = 0, off1 = 0;
	int64_t nxl = 0, nyl = 0;
	bool all_vars = false;
	int threads = 1;
	std::vector<std::vector<char>> text;  // one buffer per formatting thread
};

namespace {

// One value exactly as printf(" %.16e", v) writes it.  std::to_chars with an explicit precision is specified to produce
// printf's digits (checked against snprintf on 2e6 random bit patterns, tests/test_io_formats.py re-checks the bytes) and
// is ~3x faster; text output dominates a run on a large grid (1.6 GB per output time at 8192^2), hence the thread fan-out
// in crd_writer_write_row as well.
// The longest value is 25 characters (" -1.2345678901234567e-308": space, sign, 18 mantissa characters, 'e', sign, three
// exponent digits); buffers are sized with kMaxValueChars per value and the true end of the buffer bounds every write.
constexpr size_t kMaxValueChars = 32;
inline char *put_e16(char *out, char *end, double v)
{
	if (end - out < (ptrdiff_t)kMaxValueChars) return out;  // cannot happen with buffers sized by kMaxValueChars; never write past the end
	*out++ = ' ';
#if defined(__cpp_lib_to_chars) && __cpp_lib_to_chars >= 201611L
	if (std::isfinite(v)) {
		auto r = std::to_chars(out, end, v, std::chars_format::scientific, 16);
		if (r.ec == std::errc()) return r.ptr;
	}
#endif
	return out + std::snprintf(out, (size_t)(end - out), "%.16e", v);
}

// Host cores this process may use: affinity mask capped by the cgroup CPU quota.
int usable_threads()
{
	int n = (int)std::thread::hardware_concurrency();
	if (n < 1) n = 1;
	if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char quota[32] = {0};
		long period = 0;
		if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
			const long q = std::atol(quota) / period;
			if (q >= 1 && q < n) n = (int)q;
		}
		std::fclose(f);
	}
	if (const char *e = std::getenv("CRD_WRITER_THREADS")) {
		const int v = std::atoi(e);
		if (v >= 1) n = v;
	}
	return n > 32 ? 32 : n;
}

}  // namespace

extern "C" int crd_writer_open(const crd_run_config *cfg, const char *dir, int slab, int n_slabs, crd_writer **out)
{
	return crd_writer_open_block(cfg, dir, slab, 0, 1, slab, n_slabs, out);
}

extern "C" int crd_writer_open_block(const crd_run_config *cfg, const char *dir, int rank, int c0, int d0, int c1, int d1, crd_writer **out)
{
	if (!cfg || !out) return CRD_EINVAL;
	*out = nullptr;
	crd_grid g;
	int rc = crd_grid_from_params(&cfg->params, &g);
	if (rc != CRD_OK) return rc;
	int64_t is, ie, js, je;
	rc = crd_block_extents(g.nx, g.ny, c0, d0, c1, d1, &is, &ie, &js, &je);
	if (rc != CRD_OK) return rc;
	const int slab = rank;
	if (slab < 0 || slab > 999) return CRD_EINVAL;

	const std::string base = std::string(dir && *dir ? dir : ".") + "/" + crd::model_name(cfg->params.model) + "_" +
	                         crd::surface_name(cfg->params.surface) + "_";
	char tag[16];
	std::snprintf(tag, sizeof tag, ".%03i.txt", slab);

	FILE *fs = std::fopen((base + "subdomain" + tag).c_str(), "w");
	if (!fs) return CRD_EIO;
	std::fprintf(fs, "%li  %li  %li  %li  %li  %li %f %f %f\n", (long)g.nx, (long)g.ny, (long)is, (long)ie, (long)js, (long)je,
	             g.xmin, g.xmax, cfg->t_final);
	std::fclose(fs);

	crd_writer *w = new (std::nothrow) crd_writer;
	if (!w) return CRD_ENOMEM;
	w->nxl = ie - is + 1;
	w->nyl = je - js + 1;
	w->all_vars = (cfg->include_all_vars == 1);
	w->threads = usable_threads();
	w->f0 = ::open((base + crd::var_name(cfg->params.model, 0) + tag).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
	w->f1 = ::open((base + crd::var_name(cfg->params.model, 1) + tag).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
	if (w->f0 < 0 || w->f1 < 0) {
		crd_writer_close(w);
		return CRD_EIO;
	}
	*out = w;
	return CRD_OK;
}

extern "C" int crd_writer_write_row(crd_writer *w, const double *y_aos)
{
	if (!w || !y_aos || w->f0 < 0) return CRD_EINVAL;
	crd::TraceRange range("crd_writer_write_row");
	const int64_t n = w->nxl * w->nyl;
	const int64_t per_thread = 1 << 16;  // values formatted by one thread per round (<= 1.6 MB of text)
	const int T = w->threads;
	w->text.resize((size_t)T);
	for (auto &b : w->text) b.resize((size_t)per_thread * kMaxValueChars);
	std::vector<size_t> used((size_t)T);
	auto fan_out = [&](int active, auto &&fn) {
		std::vector<std::thread> pool;
		for (int t = 1; t < active; t++) pool.emplace_back(fn, t);
		fn(0);
		for (auto &th : pool) th.join();
	};
	auto write_all = [](int fd, const char *p, size_t len, int64_t off) {
		while (len) {
			const ssize_t k = ::pwrite(fd, p, len, (off_t)off);
			if (k <= 0) return false;
			p += k;
			len -= (size_t)k;
			off += k;
		}
		return true;
	};
	for (int var = 0; var < (w->all_vars ? 2 : 1); var++) {
		const int fd = var == 0 ? w->f0 : w->f1;
		int64_t &off = var == 0 ? w->off0 : w->off1;
		for (int64_t q0 = 0; q0 < n; q0 += per_thread * T) {
			const int active = (int)std::min<int64_t>(T, (n - q0 + per_thread - 1) / per_thread);
			fan_out(active, [&](int t) {  // phase 1: format
				const int64_t a = q0 + per_thread * t, b = std::min<int64_t>(a + per_thread, n);
				char *c = w->text[(size_t)t].data(), *end = c + w->text[(size_t)t].size();
				for (int64_t q = a; q < b; q++) c = put_e16(c, end, y_aos[2 * q + var]);
				used[(size_t)t] = (size_t)(c - w->text[(size_t)t].data());
			});
			// phase 2: append in order.  (Writes to one file serialise on the inode lock, so concurrent pwrites from the
			// formatting threads measured slower than this; the file system, ~1 GB/s, is the floor of text output.)
			for (int t = 0; t < active; t++) {
				if (!write_all(fd, w->text[(size_t)t].data(), used[(size_t)t], off)) return CRD_EIO;
				off += (int64_t)used[(size_t)t];
			}
		}
		if (!write_all(fd, "\n", 1, off)) return CRD_EIO;
		off += 1;
	}
	return CRD_OK;
}

extern "C" int crd_writer_close(crd_writer *w)
{
	if (!w) return CRD_OK;
	int rc = CRD_OK;
	if (w->f0 >= 0 && ::close(w->f0) != 0) rc = CRD_EIO;
	if (w->f1 >= 0 && ::close(w->f1) != 0) rc = CRD_EIO;
	delete w;
	return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// Binary side-channel: one NumPy .npy file per slab and variable, <Model>_<surface>_<var>.%03i.npy, holding what the
// text file holds -- one (nyl, nxl) frame per output time, shape (frames, nyl, nxl), C order, little endian -- at 8 (or 4)
// bytes per value instead of 24 characters, in the device's own layout (a field plane's owned rows are already this array).
// The header is a fixed 128-byte block rewritten at close with the number of frames actually written.
// ---------------------------------------------------------------------------------------------------------------
struct crd_npy_writer {
	int fd = -1;
	int64_t nxl = 0, nyl = 0, frames = 0, off = 128;
	int value_bytes = 8;
};

namespace {

bool npy_write_header(const crd_npy_writer *w)
{
	char head[128];
	std::memset(head, ' ', sizeof head);
	const int n = std::snprintf(head + 10, sizeof head - 10, "{'descr': '<f%d', 'fortran_order': False, 'shape': (%lld, %lld, %lld), }", w->value_bytes,
	                            (long long)w->frames, (long long)w->nyl, (long long)w->nxl);
	if (n < 0 || n >= (int)sizeof head - 11) return false;
	head[10 + n] = ' ';
	std::memcpy(head, "\x93NUMPY", 6);
	head[6] = 1;
	head[7] = 0;
	head[8] = (char)((sizeof head - 10) & 0xff);
	head[9] = (char)((sizeof head - 10) >> 8);
	head[sizeof head - 1] = '\n';
	return ::pwrite(w->fd, head, sizeof head, 0) == (ssize_t)sizeof head;
}

}  // namespace

extern "C" int crd_npy_writer_open(const crd_run_config *cfg, const char *dir, int slab, int n_slabs, int var, int value_bytes, crd_npy_writer **out)
{
	if (!cfg || !out || (var != 0 && var != 1) || (value_bytes != 4 && value_bytes != 8)) return CRD_EINVAL;
	*out = nullptr;
	crd_grid g;
	int rc = crd_grid_from_params(&cfg->params, &g);
	if (rc != CRD_OK) return rc;
	int64_t js, je;
	rc = crd_slab_extents(g.ny, slab, n_slabs, &js, &je);
	if (rc != CRD_OK) return rc;
	if (slab > 999) return CRD_EINVAL;
	char tag[16];
	std::snprintf(tag, sizeof tag, ".%03i.npy", slab);
	const std::string path = std::string(dir && *dir ? dir : ".") + "/" + crd::model_name(cfg->params.model) + "_" + crd::surface_name(cfg->params.surface) + "_" +
	                         crd::var_name(cfg->params.model, var) + tag;
	crd_npy_writer *w = new (std::nothrow) crd_npy_writer;
	if (!w) return CRD_ENOMEM;
	w->nxl = g.nx;
	w->nyl = je - js + 1;
	w->value_bytes = value_bytes;
	w->fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
	if (w->fd < 0 || !npy_write_header(w)) {
		crd_npy_writer_close(w);
		return CRD_EIO;
	}
	*out = w;
	return CRD_OK;
}

extern "C" int crd_npy_writer_append(crd_npy_writer *w, const void *frame)
{
	if (!w || !frame || w->fd < 0) return CRD_EINVAL;
	const char *p = static_cast<const char *>(frame);
	size_t len = (size_t)(w->nxl * w->nyl) * (size_t)w->value_bytes;
	while (len) {
		const ssize_t k = ::pwrite(w->fd, p, len, (off_t)w->off);
		if (k <= 0) return CRD_EIO;
		p += k;
		len -= (size_t)k;
		w->off += k;
	}
	w->frames++;
	return CRD_OK;
}

extern "C" int crd_npy_writer_close(crd_npy_writer *w)
{
	if (!w) return CRD_OK;
	int rc = CRD_OK;
	if (w->fd >= 0) {
		if (!npy_write_header(w)) rc = CRD_EIO;  // the frame count as it turned out
		if (::close(w->fd) != 0) rc = CRD_EIO;
	}
	delete w;
	return rc;
}
