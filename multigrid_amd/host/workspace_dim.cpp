// workspace_dim.cpp -- prints getBestWorkspaceDim(extents...) for one to three extents; a test driver
// for the one free function of auxiliary.h that needs no device.
#include "auxiliary.h"
#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv)
{
    if (argc < 2 || argc > 4) {
        fprintf(stderr, "usage: workspace_dim n0 [n1 [n2]]\n");
        return 2;
    }
    size_t n[3] = {0, 0, 0};
    for (int i = 1; i < argc; ++i)
        n[i - 1] = (size_t)atol(argv[i]);
    const mgrt::NDRange in = argc == 2 ? mgrt::NDRange(n[0]) : argc == 3 ? mgrt::NDRange(n[0], n[1]) : mgrt::NDRange(n[0], n[1], n[2]);
    const mgrt::NDRange out = getBestWorkspaceDim(in);
    for (size_t i = 0; i < out.dimensions(); ++i)
        printf("%s%zu", i ? " " : "", out[i]);
    printf("\n");
    return 0;
}
