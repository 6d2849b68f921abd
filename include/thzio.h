/* thzio.h — C ABI of libthzio.so: the dotTHz (.thz / .thzimg, HDF5) reader and writer
 * that feeds the recompute path.
 *
 * Replaces, for the C++ host mirror and the tools of this repo, what the reference
 * reaches through the `dotthz` crate (0.3.0, Cargo.toml:64 / Cargo.lock:2974; absent from /root/reference —
 * the published dotTHz layout is restated here: one HDF5 group per measurement with the
 * string attributes thzVer, dsDescription, mdDescription, md1..mdN, user, date, time,
 * mode, instrument and the datasets ds1..dsN):
 *   open_scan_from_thz   src/io.rs:496-631   -> thz_io_open / thz_io_shape / thz_io_read_*
 *   open_pulse_from_thz  src/io.rs:435-477   -> thz_io_read_pulse
 *   save_to_thz          src/io.rs:405-432   -> thz_io_save_scan
 * The Rust application itself keeps its dotthz reader and hands the cube to
 * thz_session_upload (include/thzgpu.h); see INTEGRATION.md.
 *
 * Kept out of libthzgpu.so so that the engine has no HDF5 dependency.  Not thread-safe
 * (HDF5 is not); one file handle per call site.
 */
#ifndef THZIO_H
#define THZIO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct thz_io_file thz_io_file;

enum {
    THZ_IO_OK = 0,
    THZ_IO_ERR_INVALID = -1,
    THZ_IO_ERR_OPEN = -2,    /* file missing / not HDF5 */
    THZ_IO_ERR_FORMAT = -3,  /* no group / no usable dataset */
    THZ_IO_ERR_HDF5 = -4
};

/* message of the last failing call on this thread */
const char *thz_io_last_error(void);

/* Opens the file and selects its first group in name order (io.rs:507-518: "opening only
 * the first"). */
int thz_io_open(const char *path, thz_io_file **out);
void thz_io_close(thz_io_file *f);
/* number of groups in the file and the name of the selected one */
size_t thz_io_group_count(const thz_io_file *f);
const char *thz_io_group_name(const thz_io_file *f);

/* Dataset search of open_scan_from_thz (io.rs:523-566): time = the first 1-D dataset,
 * cube = the first 3-D dataset; if neither exists, the first dataset read as a 2-D
 * (nt, 2) single pulse -> (1, 1, nt) with time = column 0.  kind: 0 scan, 1 single pulse. */
int thz_io_shape(thz_io_file *f, size_t *nx, size_t *ny, size_t *nt, int *kind);
int thz_io_read_time(thz_io_file *f, float *time /* nt */);
/* x-rows [x0, x0 + n) of the cube, C order (n, ny, nt) — one hyperslab per call, so a
 * loader can stream slabs to the device while the next one is read */
int thz_io_read_cube(thz_io_file *f, size_t x0, size_t n, float *dst);

/* Metadata map of the selected group: mdDescription names -> md1..mdN values as text
 * (numbers are printed the way Rust's to_string prints them).  Returns the value's
 * length, or -1 when the key is absent; buf may be NULL. */
long thz_io_metadata(thz_io_file *f, const char *key, char *buf, size_t cap);
/* plain string attributes of the group: "dsDescription", "mdDescription", "thzVer", ... */
long thz_io_attribute(thz_io_file *f, const char *name, char *buf, size_t cap);

/* Geometry overrides of io.rs:569-613: width / height from the metadata when they parse as
 * unsigned integers (else the cube's own shape), dx, dy, x_min, y_min when they parse as
 * f32 (has_* = 0 otherwise). */
typedef struct thz_io_geometry {
    size_t width, height;
    float dx, dy, x_min, y_min;
    int32_t has_dx, has_dy, has_x_min, has_y_min;
} thz_io_geometry;
int thz_io_get_geometry(thz_io_file *f, thz_io_geometry *out);

/* open_pulse_from_thz (io.rs:435-477): first dataset of the first group as 2-D (n, 2):
 * column 0 = time, column 1 = signal.  Sizing call with time = signal = NULL. */
int thz_io_read_pulse(const char *path, size_t *n, float *time, float *signal);

/* save_to_thz (io.rs:405-432): group "Image", ds1 = time, ds2 = cube ("time, dataset"),
 * metadata written as variable-length UTF-8 string attributes. */
int thz_io_save_scan(const char *path, const float *time, size_t nt, const float *cube, size_t nx, size_t ny,
                     const char *const *md_keys, const char *const *md_values, size_t n_md);
/* a single-pulse file: group `group`, ds1 = (n, 2) [time, signal] */
int thz_io_save_pulse(const char *path, const char *group, const float *time, const float *signal, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* THZIO_H */
