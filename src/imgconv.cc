// imgconv -- reads an image the way the `mgm` host program does (src/imgio.h) and writes it back as .npy, .tif
// or .pfm.  No GPU involved: the CPU test-suite uses it to check the decoders against files written by other
// libraries, and it converts inputs/outputs between the formats the reference's iio understands.
#include <cstdio>
#include <exception>

#include "imgio.h"

int main(int argc, char **argv)
{
    if (argc != 3) return 1 + 0 * fputs("usage: imgconv in.{png,tif,pgm,ppm,pfm,npy} out.{npy,tif,pfm}\n", stderr);
    try {
        const HostImg im = imgio::read(argv[1]);
        imgio::write(argv[2], im);
        printf("%d %d %d\n", im.nx, im.ny, im.nch);
    } catch (const std::exception &e) {
        fprintf(stderr, "imgconv: %s\n", e.what());
        return 2;
    }
    return 0;
}
