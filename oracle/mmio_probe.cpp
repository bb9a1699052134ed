// mmio_probe.cpp — runs the REFERENCE's own MatrixMarket banner / size-line parser
// (util/mmio.hpp:215-298 mm_read_banner, 308-336 mm_read_mtx_crd_size), compiled
// from the reference checkout where it lies (-I/root/reference; nothing is copied
// into this repository). Test infrastructure: used by tests/ to validate that the
// oracle's and the product loader's header handling agree with the reference on
// the bundled matrices and on hand-made edge-case files.
//
// usage: mmio_probe file.mtx  ->  one line:
//   "<banner_rc> <size_rc> <M> <N> <nz> <typecode 4 chars>"
#include <cstdio>
#include <cstdlib>

#include "util/mmio.hpp"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "r");
    if (!f) {
        printf("nofile\n");
        return 1;
    }
    MM_typecode tc;
    int brc = mm_read_banner(f, &tc);
    int M = 0, N = 0, nz = 0, src = -1;
    if (brc == 0) src = mm_read_mtx_crd_size(f, &M, &N, &nz);
    printf("%d %d %d %d %d %c%c%c%c\n", brc, src, M, N, nz, brc == 0 ? tc[0] : '-', brc == 0 ? tc[1] : '-',
           brc == 0 ? tc[2] : '-', brc == 0 ? tc[3] : '-');
    fclose(f);
    return 0;
}
