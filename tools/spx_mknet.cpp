// spx_mknet: write the repo's synthetic CBNF network to a file.
//   usage: spx_mknet <seed> <preset 0|1|2|3> <out.nnue>
// Used by oracle/Makefile to give the compiled reference an embeddable net (the default net is not obtainable
// offline); the same bytes are generated in-process by spx_synth_net() on the GPU box.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../stormphrax_amd/csrc/spx_internal.h"

int main(int argc, char** argv) {
    if (argc != 4) {
        std::fprintf(stderr, "usage: %s <seed> <preset> <out>\n", argv[0]);
        return 1;
    }
    const uint64_t seed = std::strtoull(argv[1], nullptr, 10);
    const int preset = std::atoi(argv[2]);
    std::vector<unsigned char> buf(spx::synthNetBytes());
    if (!spx::synthNet(seed, preset, buf.data(), buf.size())) {
        std::fprintf(stderr, "bad arguments\n");
        return 1;
    }
    FILE* f = std::fopen(argv[3], "wb");
    if (!f || std::fwrite(buf.data(), 1, buf.size(), f) != buf.size()) {
        std::fprintf(stderr, "write failed\n");
        return 1;
    }
    std::fclose(f);
    std::printf("%s: %zu bytes, fnv1a64 %016llx\n", argv[3], buf.size(),
                static_cast<unsigned long long>(spx::fnv1a64(buf.data(), buf.size())));
    return 0;
}
