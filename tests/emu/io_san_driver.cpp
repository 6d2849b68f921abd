// io_san_driver.cpp — the dotTHz reader / writer (thz_image_explorer_amd/io/thz_io.cpp) under
// AddressSanitizer + UBSan: a real sample file, a written-and-reread scan, tiny caller buffers, a missing
// file and a file that is not HDF5.  TEST INFRASTRUCTURE ONLY; argv[1] = sample .thz, argv[2] = scratch dir.
#include "thzio.h"

#include <hdf5.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const std::string sample = argv[1], dir = argv[2];
    thz_io_file *f = nullptr;
    if (thz_io_open(sample.c_str(), &f) != 0) { std::printf("open failed: %s\n", thz_io_last_error()); return 1; }
    size_t nx = 0, ny = 0, nt = 0;
    int kind = -1;
    if (thz_io_shape(f, &nx, &ny, &nt, &kind) != 0) return 1;
    std::vector<float> t(nt), cube(nx * ny * nt);
    if (thz_io_read_time(f, t.data()) != 0 || thz_io_read_cube(f, 0, nx, cube.data()) != 0) return 1;
    (void)thz_io_read_cube(f, nx, 1, cube.data());   // past the end: an error, not a write
    (void)thz_io_group_count(f);
    (void)thz_io_group_name(f);
    char small[3], big[4096];
    for (const char *key : {"dx", "dy", "width", "nonexistent", ""}) {
        (void)thz_io_metadata(f, key, nullptr, 0);
        (void)thz_io_metadata(f, key, small, sizeof small);
        (void)thz_io_metadata(f, key, big, sizeof big);
    }
    for (const char *a : {"mdDescription", "dsDescription", "thzVer", "nope"}) {
        (void)thz_io_attribute(f, a, small, sizeof small);
        (void)thz_io_attribute(f, a, big, sizeof big);
    }
    thz_io_geometry g{};
    (void)thz_io_get_geometry(f, &g);
    thz_io_close(f);
    size_t n = 0;
    if (thz_io_read_pulse(sample.c_str(), &n, nullptr, nullptr) == 0) {
        std::vector<float> pt(n), ps(n);
        (void)thz_io_read_pulse(sample.c_str(), &n, pt.data(), ps.data());
    }
    // write a scan and read it back
    {
        const size_t wx = 3, wy = 2, wt = 17;
        std::vector<float> tm(wt), cb(wx * wy * wt);
        for (size_t i = 0; i < wt; ++i) tm[i] = 0.05f * (float)i;
        for (size_t i = 0; i < cb.size(); ++i) cb[i] = (float)i;
        const char *keys[] = {"dx", "dy", "width", "height", "user"};
        const char *vals[] = {"0.5", "0.25", "3", "2", "x"};
        const std::string p = dir + "/san_scan.thz";
        if (thz_io_save_scan(p.c_str(), tm.data(), wt, cb.data(), wx, wy, keys, vals, 5) != 0) return 1;
        thz_io_file *h = nullptr;
        if (thz_io_open(p.c_str(), &h) != 0) return 1;
        size_t a = 0, b = 0, c = 0;
        int k = 0;
        if (thz_io_shape(h, &a, &b, &c, &k) != 0 || a != wx || b != wy || c != wt) return 1;
        std::vector<float> slab(wy * wt);
        for (size_t x = 0; x < wx; ++x)
            if (thz_io_read_cube(h, x, 1, slab.data()) != 0 || slab[0] != cb[x * wy * wt]) return 1;
        (void)thz_io_get_geometry(h, &g);
        thz_io_close(h);
        const std::string q = dir + "/san_pulse.thz";
        if (thz_io_save_pulse(q.c_str(), "ref", tm.data(), tm.data(), wt) != 0) return 1;
    }
    // not there / not HDF5
    thz_io_file *bad = nullptr;
    (void)thz_io_open((dir + "/does_not_exist.thz").c_str(), &bad);
    {
        const std::string p = dir + "/garbage.thz";
        FILE *fp = std::fopen(p.c_str(), "wb");
        if (fp) { std::fputs("this is not an HDF5 file", fp); std::fclose(fp); }
        (void)thz_io_open(p.c_str(), &bad);
        (void)thz_io_read_pulse(p.c_str(), &n, nullptr, nullptr);
    }
    // files that are not what their shape says: a time axis longer than the cube's traces must be
    // refused (callers size the time buffer from the cube), and a dataset hdf5-rust could not read as f32 (a
    // string) in front of the time axis is skipped like the reference's role loop skips it (io.rs:523-566)
    {
        const std::string p = dir + "/mismatch.thz";
        hid_t file = H5Fcreate(p.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
        hid_t grp = H5Gcreate2(file, "scan", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        const hsize_t n1[1] = {100000}, n3[3] = {2, 2, 8};
        std::vector<float> big1(100000, 1.0f), cube3(32, 2.0f);
        hid_t s1 = H5Screate_simple(1, n1, nullptr), s3 = H5Screate_simple(3, n3, nullptr);
        hid_t d1 = H5Dcreate2(grp, "ds1", H5T_NATIVE_FLOAT, s1, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        hid_t d3 = H5Dcreate2(grp, "ds2", H5T_NATIVE_FLOAT, s3, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Dwrite(d1, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, big1.data());
        H5Dwrite(d3, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, cube3.data());
        H5Dclose(d1); H5Dclose(d3); H5Sclose(s1); H5Sclose(s3); H5Gclose(grp); H5Fclose(file);
        thz_io_file *h = nullptr;
        if (thz_io_open(p.c_str(), &h) != THZ_IO_ERR_FORMAT || h != nullptr) { std::printf("mismatched file was accepted\n"); return 1; }
    }
    {
        const std::string p = dir + "/stringfirst.thz";
        hid_t file = H5Fcreate(p.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
        hid_t grp = H5Gcreate2(file, "scan", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        const hsize_t n1[1] = {8}, n3[3] = {2, 2, 8}, ns[1] = {3};
        hid_t st = H5Tcopy(H5T_C_S1);
        H5Tset_size(st, 4);
        const char labels[12] = {'a', 'b', 'c', 0, 'd', 'e', 'f', 0, 'g', 'h', 'i', 0};
        std::vector<double> tax(8);
        std::vector<int> cube3(32);
        for (int i = 0; i < 8; ++i) tax[i] = 0.5 * i;
        for (int i = 0; i < 32; ++i) cube3[i] = i;
        hid_t ss = H5Screate_simple(1, ns, nullptr), s1 = H5Screate_simple(1, n1, nullptr), s3 = H5Screate_simple(3, n3, nullptr);
        hid_t d0 = H5Dcreate2(grp, "a_label", st, ss, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        hid_t d1 = H5Dcreate2(grp, "b_time", H5T_NATIVE_DOUBLE, s1, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        hid_t d3 = H5Dcreate2(grp, "c_cube", H5T_NATIVE_INT, s3, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Dwrite(d0, st, H5S_ALL, H5S_ALL, H5P_DEFAULT, labels);
        H5Dwrite(d1, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, tax.data());
        H5Dwrite(d3, H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, cube3.data());
        H5Dclose(d0); H5Dclose(d1); H5Dclose(d3); H5Sclose(ss); H5Sclose(s1); H5Sclose(s3); H5Tclose(st); H5Gclose(grp); H5Fclose(file);
        thz_io_file *h = nullptr;
        if (thz_io_open(p.c_str(), &h) != 0) { std::printf("string-first file refused: %s\n", thz_io_last_error()); return 1; }
        size_t a = 0, b = 0, c = 0;
        int k = -1;
        if (thz_io_shape(h, &a, &b, &c, &k) != 0 || a != 2 || b != 2 || c != 8 || k != 0) return 1;
        std::vector<float> tt(8), cc(32);
        if (thz_io_read_time(h, tt.data()) != 0 || tt[3] != 1.5f) return 1;
        if (thz_io_read_cube(h, 0, 2, cc.data()) != 0 || cc[31] != 31.0f) return 1;
        thz_io_close(h);
    }
    std::printf("io done\n");
    return 0;
}
