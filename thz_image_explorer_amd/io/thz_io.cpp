// thz_io.cpp — dotTHz reader / writer over the HDF5 C API (include/thzio.h).
#include "../../include/thzio.h"

#include <hdf5.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

struct Dataset {
    std::string name;
    int ndim = 0;
    hsize_t dims[3] = {0, 0, 0};
    // hdf5-rust's read_1d::<f32>() / read_dyn::<f32>() (io.rs:523-566) succeed for every dataset HDF5 can
    // convert to a native float — integers and floats of any width (Conversion::Soft, the reader's default)
    // — and fail for strings, compounds, references ...: the reference's role loop then tries the next dataset
    bool numeric = false;
};

// Rust's f64::to_string: integers without a fraction, otherwise the shortest digits that round-trip
std::string rust_float_string(double v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    char buf[64];
    if (v == std::floor(v) && std::fabs(v) < 1e16) {
        std::snprintf(buf, sizeof buf, "%.0f", v);
        return buf;
    }
    for (int prec = 1; prec <= 17; ++prec) {
        std::snprintf(buf, sizeof buf, "%.*g", prec, v);
        if (std::strtod(buf, nullptr) == v) break;
    }
    return buf;
}

std::vector<std::string> split_list(const std::string &s)
{
    std::vector<std::string> out;
    size_t pos = 0;
    while (pos <= s.size()) {
        size_t c = s.find(',', pos);
        if (c == std::string::npos) c = s.size();
        std::string item = s.substr(pos, c - pos);
        const size_t a = item.find_first_not_of(" \t");
        const size_t b = item.find_last_not_of(" \t");
        out.push_back(a == std::string::npos ? std::string() : item.substr(a, b - a + 1));
        pos = c + 1;
    }
    return out;
}

// an attribute of any supported type as text; false if absent / unsupported
bool read_attr_text(hid_t obj, const char *name, std::string &out)
{
    if (H5Aexists(obj, name) <= 0) return false;
    hid_t a = H5Aopen(obj, name, H5P_DEFAULT);
    if (a < 0) return false;
    hid_t t = H5Aget_type(a);
    bool ok = false;
    const H5T_class_t cls = H5Tget_class(t);
    if (cls == H5T_STRING) {
        if (H5Tis_variable_str(t) > 0) {
            char *p = nullptr;
            hid_t mt = H5Tcopy(H5T_C_S1);
            H5Tset_size(mt, H5T_VARIABLE);
            H5Tset_cset(mt, H5Tget_cset(t));
            if (H5Aread(a, mt, &p) >= 0) {
                out = p ? p : "";
                ok = true;
                if (p) H5free_memory(p);
            }
            H5Tclose(mt);
        } else {
            const size_t n = H5Tget_size(t);
            std::vector<char> buf(n + 1, 0);
            if (H5Aread(a, t, buf.data()) >= 0) {
                out = std::string(buf.data(), strnlen(buf.data(), n));
                ok = true;
            }
        }
    } else if (cls == H5T_FLOAT) {
        double v = 0;
        if (H5Aread(a, H5T_NATIVE_DOUBLE, &v) >= 0) {
            out = rust_float_string(v);
            ok = true;
        }
    } else if (cls == H5T_INTEGER) {
        long long v = 0;
        if (H5Aread(a, H5T_NATIVE_LLONG, &v) >= 0) {
            out = std::to_string(v);
            ok = true;
        }
    }
    H5Tclose(t);
    H5Aclose(a);
    return ok;
}

herr_t collect_names(hid_t, const char *name, const H5L_info_t *, void *op)
{
    static_cast<std::vector<std::string> *>(op)->push_back(name);
    return 0;
}

bool write_str_attr(hid_t obj, const char *name, const std::string &value)
{
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, H5T_VARIABLE);
    H5Tset_cset(t, H5T_CSET_UTF8);
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(obj, name, t, sp, H5P_DEFAULT, H5P_DEFAULT);
    const char *p = value.c_str();
    const bool ok = a >= 0 && H5Awrite(a, t, &p) >= 0;
    if (a >= 0) H5Aclose(a);
    H5Sclose(sp);
    H5Tclose(t);
    return ok;
}

bool write_dataset(hid_t group, const char *name, int ndim, const hsize_t *dims, const float *data)
{
    hid_t sp = H5Screate_simple(ndim, dims, nullptr);
    hid_t d = H5Dcreate2(group, name, H5T_IEEE_F32LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    const bool ok = d >= 0 && H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) >= 0;
    if (d >= 0) H5Dclose(d);
    H5Sclose(sp);
    return ok;
}

struct QuietHdf5 {
    QuietHdf5() { H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr); }
};

}  // namespace

struct thz_io_file {
    hid_t file = -1, group = -1;
    std::vector<std::string> groups;
    std::string group_name;
    std::vector<Dataset> datasets;  // name order, like hdf5-rust's Group::datasets()
    std::map<std::string, std::string> md;
    int kind = -1;  // 0 scan, 1 single pulse
    int time_ds = -1, cube_ds = -1;
    size_t nx = 0, ny = 0, nt = 0;
};

extern "C" {

const char *thz_io_last_error(void) { return g_err.c_str(); }

int thz_io_open(const char *path, thz_io_file **out)
{
    static QuietHdf5 quiet;
    if (!path || !out) return fail(THZ_IO_ERR_INVALID, "thz_io_open: null argument");
    *out = nullptr;
    hid_t file = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (file < 0) return fail(THZ_IO_ERR_OPEN, std::string("cannot open ") + path);
    thz_io_file *f = new thz_io_file();
    f->file = file;
    std::vector<std::string> names;
    H5Literate(file, H5_INDEX_NAME, H5_ITER_INC, nullptr, collect_names, &names);
    for (const auto &n : names) {
        H5O_info_t info;
        if (H5Oget_info_by_name(file, n.c_str(), &info, H5P_DEFAULT) >= 0 && info.type == H5O_TYPE_GROUP)
            f->groups.push_back(n);
    }
    if (f->groups.empty()) {
        thz_io_close(f);
        return fail(THZ_IO_ERR_FORMAT, std::string(path) + ": no group");
    }
    f->group_name = f->groups.front();
    f->group = H5Gopen2(file, f->group_name.c_str(), H5P_DEFAULT);
    if (f->group < 0) {
        thz_io_close(f);
        return fail(THZ_IO_ERR_HDF5, "cannot open group " + f->group_name);
    }
    names.clear();
    H5Literate(f->group, H5_INDEX_NAME, H5_ITER_INC, nullptr, collect_names, &names);
    for (const auto &n : names) {
        H5O_info_t info;
        if (H5Oget_info_by_name(f->group, n.c_str(), &info, H5P_DEFAULT) < 0 || info.type != H5O_TYPE_DATASET) continue;
        hid_t d = H5Dopen2(f->group, n.c_str(), H5P_DEFAULT);
        if (d < 0) continue;
        hid_t sp = H5Dget_space(d);
        Dataset ds;
        ds.name = n;
        ds.ndim = H5Sget_simple_extent_ndims(sp);
        if (ds.ndim >= 1 && ds.ndim <= 3) H5Sget_simple_extent_dims(sp, ds.dims, nullptr);
        H5Sclose(sp);
        const hid_t ty = H5Dget_type(d);
        if (ty >= 0) {
            const H5T_class_t cls = H5Tget_class(ty);
            ds.numeric = cls == H5T_FLOAT || cls == H5T_INTEGER;
            H5Tclose(ty);
        }
        H5Dclose(d);
        f->datasets.push_back(ds);
    }
    // metadata map: mdDescription[i] -> md{i+1}
    std::string desc;
    if (read_attr_text(f->group, "mdDescription", desc)) {
        const auto keys = split_list(desc);
        for (size_t i = 0; i < keys.size(); ++i) {
            std::string v;
            if (read_attr_text(f->group, ("md" + std::to_string(i + 1)).c_str(), v)) f->md[keys[i]] = v;
        }
    }
    // dataset roles, io.rs:523-566
    for (size_t i = 0; i < f->datasets.size(); ++i)
        if (f->datasets[i].ndim == 1 && f->datasets[i].numeric) { f->time_ds = (int)i; break; }
    for (size_t i = 0; i < f->datasets.size(); ++i)
        if (f->datasets[i].ndim == 3 && f->datasets[i].numeric) { f->cube_ds = (int)i; break; }
    // A time axis that is not as long as the cube's traces is not a scan any stage could process (the
    // reference would hand mismatched arrays to ndarray's Zip and panic); callers size their time buffer
    // from the cube's last dimension, so such a file is refused here rather than read past that buffer.
    if (f->time_ds >= 0 && f->cube_ds >= 0 && f->datasets[f->time_ds].dims[0] != f->datasets[f->cube_ds].dims[2]) {
        const std::string msg = "group " + f->group_name + ": time axis has " + std::to_string(f->datasets[f->time_ds].dims[0])
                                + " samples, the cube's traces " + std::to_string(f->datasets[f->cube_ds].dims[2]);
        thz_io_close(f);
        return fail(THZ_IO_ERR_FORMAT, msg);
    }
    if (f->time_ds >= 0 || f->cube_ds >= 0) {
        f->kind = 0;
        if (f->cube_ds >= 0) {
            f->nx = f->datasets[f->cube_ds].dims[0];
            f->ny = f->datasets[f->cube_ds].dims[1];
            f->nt = f->datasets[f->cube_ds].dims[2];
        }
        if (f->time_ds >= 0 && f->cube_ds < 0) f->nt = f->datasets[f->time_ds].dims[0];
    } else if (!f->datasets.empty() && f->datasets[0].ndim == 2 && f->datasets[0].dims[1] >= 2 && f->datasets[0].numeric) {
        f->kind = 1;
        f->nx = f->ny = 1;
        f->nt = f->datasets[0].dims[0];
    }
    *out = f;
    return THZ_IO_OK;
}

void thz_io_close(thz_io_file *f)
{
    if (!f) return;
    if (f->group >= 0) H5Gclose(f->group);
    if (f->file >= 0) H5Fclose(f->file);
    delete f;
}

size_t thz_io_group_count(const thz_io_file *f) { return f ? f->groups.size() : 0; }
const char *thz_io_group_name(const thz_io_file *f) { return f ? f->group_name.c_str() : ""; }

int thz_io_shape(thz_io_file *f, size_t *nx, size_t *ny, size_t *nt, int *kind)
{
    if (!f) return fail(THZ_IO_ERR_INVALID, "thz_io_shape: null file");
    if (f->kind < 0) return fail(THZ_IO_ERR_FORMAT, "group " + f->group_name + ": no scan or pulse dataset");
    if (nx) *nx = f->nx;
    if (ny) *ny = f->ny;
    if (nt) *nt = f->nt;
    if (kind) *kind = f->kind;
    return THZ_IO_OK;
}

int thz_io_read_time(thz_io_file *f, float *time)
{
    if (!f || !time) return fail(THZ_IO_ERR_INVALID, "thz_io_read_time: null argument");
    if (f->kind == 0) {
        if (f->time_ds < 0) return fail(THZ_IO_ERR_FORMAT, "no 1-D time dataset");
        hid_t d = H5Dopen2(f->group, f->datasets[f->time_ds].name.c_str(), H5P_DEFAULT);
        if (d < 0) return fail(THZ_IO_ERR_HDF5, "cannot open the time dataset");
        // exactly f->nt elements, whatever the dataset's extent says now: the caller's buffer has that many
        const hsize_t start[1] = {0}, count[1] = {(hsize_t)f->nt};
        hid_t fs = H5Dget_space(d);
        herr_t rc = -1;
        if (fs >= 0 && H5Sget_simple_extent_ndims(fs) == 1 && H5Sselect_hyperslab(fs, H5S_SELECT_SET, start, nullptr, count, nullptr) >= 0) {
            hid_t ms = H5Screate_simple(1, count, nullptr);
            if (ms >= 0) {
                rc = H5Dread(d, H5T_NATIVE_FLOAT, ms, fs, H5P_DEFAULT, time);
                H5Sclose(ms);
            }
        }
        if (fs >= 0) H5Sclose(fs);
        H5Dclose(d);
        return rc < 0 ? fail(THZ_IO_ERR_HDF5, "reading the time dataset failed") : THZ_IO_OK;
    }
    if (f->kind == 1) {
        const Dataset &ds = f->datasets[0];
        std::vector<float> buf((size_t)ds.dims[0] * ds.dims[1]);
        hid_t d = H5Dopen2(f->group, ds.name.c_str(), H5P_DEFAULT);
        if (d < 0) return fail(THZ_IO_ERR_HDF5, "cannot open the pulse dataset");
        const herr_t rc = H5Dread(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.data());
        H5Dclose(d);
        if (rc < 0) return fail(THZ_IO_ERR_HDF5, "reading the pulse dataset failed");
        for (size_t i = 0; i < (size_t)ds.dims[0]; ++i) time[i] = buf[i * ds.dims[1]];
        return THZ_IO_OK;
    }
    return fail(THZ_IO_ERR_FORMAT, "no dataset");
}

int thz_io_read_cube(thz_io_file *f, size_t x0, size_t n, float *dst)
{
    if (!f || !dst) return fail(THZ_IO_ERR_INVALID, "thz_io_read_cube: null argument");
    if (x0 + n > f->nx) return fail(THZ_IO_ERR_INVALID, "thz_io_read_cube: rows out of range");
    if (n == 0) return THZ_IO_OK;
    if (f->kind == 1) {
        const Dataset &ds = f->datasets[0];
        std::vector<float> buf((size_t)ds.dims[0] * ds.dims[1]);
        hid_t d = H5Dopen2(f->group, ds.name.c_str(), H5P_DEFAULT);
        if (d < 0) return fail(THZ_IO_ERR_HDF5, "cannot open the pulse dataset");
        const herr_t rc = H5Dread(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.data());
        H5Dclose(d);
        if (rc < 0) return fail(THZ_IO_ERR_HDF5, "reading the pulse dataset failed");
        for (size_t i = 0; i < (size_t)ds.dims[0]; ++i) dst[i] = buf[i * ds.dims[1] + 1];
        return THZ_IO_OK;
    }
    if (f->kind != 0 || f->cube_ds < 0) return fail(THZ_IO_ERR_FORMAT, "no 3-D dataset");
    hid_t d = H5Dopen2(f->group, f->datasets[f->cube_ds].name.c_str(), H5P_DEFAULT);
    if (d < 0) return fail(THZ_IO_ERR_HDF5, "cannot open the cube dataset");
    hid_t fs = H5Dget_space(d);
    const hsize_t start[3] = {x0, 0, 0}, count[3] = {n, f->ny, f->nt};
    H5Sselect_hyperslab(fs, H5S_SELECT_SET, start, nullptr, count, nullptr);
    hid_t ms = H5Screate_simple(3, count, nullptr);
    const herr_t rc = H5Dread(d, H5T_NATIVE_FLOAT, ms, fs, H5P_DEFAULT, dst);
    H5Sclose(ms);
    H5Sclose(fs);
    H5Dclose(d);
    return rc < 0 ? fail(THZ_IO_ERR_HDF5, "reading the cube failed") : THZ_IO_OK;
}

static long copy_out(const std::string &v, char *buf, size_t cap)
{
    if (buf && cap) {
        const size_t n = std::min(cap - 1, v.size());
        std::memcpy(buf, v.data(), n);
        buf[n] = 0;
    }
    return (long)v.size();
}

long thz_io_metadata(thz_io_file *f, const char *key, char *buf, size_t cap)
{
    if (!f || !key) return -1;
    const auto it = f->md.find(key);
    if (it == f->md.end()) return -1;
    return copy_out(it->second, buf, cap);
}

long thz_io_attribute(thz_io_file *f, const char *name, char *buf, size_t cap)
{
    if (!f || !name) return -1;
    std::string v;
    if (!read_attr_text(f->group, name, v)) return -1;
    return copy_out(v, buf, cap);
}

// Rust's str::parse::<usize>: optional '+', digits only
static bool parse_usize(const std::string &s, size_t *out)
{
    size_t i = 0;
    if (i < s.size() && s[i] == '+') ++i;
    if (i == s.size()) return false;
    size_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (size_t)(s[i] - '0');
    }
    *out = v;
    return true;
}

static bool parse_f32(const std::string &s, float *out)
{
    if (s.empty() || s.front() == ' ' || s.back() == ' ') return false;
    char *end = nullptr;
    const float v = std::strtof(s.c_str(), &end);
    if (end != s.c_str() + s.size()) return false;
    *out = v;
    return true;
}

int thz_io_get_geometry(thz_io_file *f, thz_io_geometry *g)
{
    if (!f || !g) return fail(THZ_IO_ERR_INVALID, "thz_io_get_geometry: null argument");
    std::memset(g, 0, sizeof *g);
    g->width = f->nx;   // scan.width / height default to the cube's shape (single pulse: 1, 1)
    g->height = f->ny;
    if (f->kind == 1) { g->dx = g->dy = 1.0f; g->has_dx = g->has_dy = 1; }   // io.rs:558-559
    auto md = [&](const char *k, std::string &v) {
        const auto it = f->md.find(k);
        if (it == f->md.end()) return false;
        v = it->second;
        return true;
    };
    std::string v;
    size_t u;
    if (md("width", v) && parse_usize(v, &u)) g->width = u;
    if (md("height", v) && parse_usize(v, &u)) g->height = u;
    float x;
    if (md("dx [mm]", v)) { g->has_dx = parse_f32(v, &x); g->dx = g->has_dx ? x : 0.0f; }
    if (md("dy [mm]", v)) { g->has_dy = parse_f32(v, &x); g->dy = g->has_dy ? x : 0.0f; }
    if (md("x_min [mm]", v)) { g->has_x_min = parse_f32(v, &x); g->x_min = g->has_x_min ? x : 0.0f; }
    if (md("y_min [mm]", v)) { g->has_y_min = parse_f32(v, &x); g->y_min = g->has_y_min ? x : 0.0f; }
    return THZ_IO_OK;
}

int thz_io_read_pulse(const char *path, size_t *n, float *time, float *signal)
{
    if (!n) return fail(THZ_IO_ERR_INVALID, "thz_io_read_pulse: null argument");
    thz_io_file *f = nullptr;
    if (int rc = thz_io_open(path, &f)) return rc;
    int rc = THZ_IO_OK;
    if (f->datasets.empty() || f->datasets[0].ndim != 2 || f->datasets[0].dims[1] < 2) {
        // the reference returns empty vectors here (io.rs:441-442, 466-471)
        *n = 0;
    } else {
        const Dataset &ds = f->datasets[0];
        *n = (size_t)ds.dims[0];
        if (time || signal) {
            std::vector<float> buf((size_t)ds.dims[0] * ds.dims[1]);
            hid_t d = H5Dopen2(f->group, ds.name.c_str(), H5P_DEFAULT);
            if (H5Dread(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.data()) < 0)
                rc = fail(THZ_IO_ERR_HDF5, "reading the pulse dataset failed");
            H5Dclose(d);
            for (size_t i = 0; !rc && i < *n; ++i) {
                if (time) time[i] = buf[i * ds.dims[1]];
                if (signal) signal[i] = buf[i * ds.dims[1] + 1];
            }
        }
    }
    thz_io_close(f);
    return rc;
}

int thz_io_save_scan(const char *path, const float *time, size_t nt, const float *cube, size_t nx, size_t ny,
                     const char *const *md_keys, const char *const *md_values, size_t n_md)
{
    static QuietHdf5 quiet;
    if (!path || !time || !cube || (n_md && (!md_keys || !md_values)))
        return fail(THZ_IO_ERR_INVALID, "thz_io_save_scan: null argument");
    hid_t file = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (file < 0) return fail(THZ_IO_ERR_OPEN, std::string("cannot create ") + path);
    hid_t g = H5Gcreate2(file, "Image", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    bool ok = g >= 0;
    ok = ok && write_str_attr(g, "thzVer", "1.00");
    ok = ok && write_str_attr(g, "dsDescription", "time, dataset");
    std::string desc;
    for (size_t i = 0; i < n_md; ++i) desc += (i ? ", " : "") + std::string(md_keys[i]);
    ok = ok && write_str_attr(g, "mdDescription", desc);
    for (size_t i = 0; ok && i < n_md; ++i) ok = write_str_attr(g, ("md" + std::to_string(i + 1)).c_str(), md_values[i]);
    const hsize_t d1[1] = {nt}, d3[3] = {nx, ny, nt};
    ok = ok && write_dataset(g, "ds1", 1, d1, time);
    ok = ok && write_dataset(g, "ds2", 3, d3, cube);
    if (g >= 0) H5Gclose(g);
    H5Fclose(file);
    return ok ? THZ_IO_OK : fail(THZ_IO_ERR_HDF5, std::string("writing ") + path + " failed");
}

int thz_io_save_pulse(const char *path, const char *group, const float *time, const float *signal, size_t n)
{
    static QuietHdf5 quiet;
    if (!path || !group || !time || !signal) return fail(THZ_IO_ERR_INVALID, "thz_io_save_pulse: null argument");
    hid_t file = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (file < 0) return fail(THZ_IO_ERR_OPEN, std::string("cannot create ") + path);
    hid_t g = H5Gcreate2(file, group, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    std::vector<float> buf(2 * n);
    for (size_t i = 0; i < n; ++i) {
        buf[2 * i] = time[i];
        buf[2 * i + 1] = signal[i];
    }
    const hsize_t d2[2] = {n, 2};
    bool ok = g >= 0 && write_str_attr(g, "thzVer", "1.00") && write_str_attr(g, "dsDescription", "time, signal")
              && write_dataset(g, "ds1", 2, d2, buf.data());
    if (g >= 0) H5Gclose(g);
    H5Fclose(file);
    return ok ? THZ_IO_OK : fail(THZ_IO_ERR_HDF5, std::string("writing ") + path + " failed");
}

}  // extern "C"
