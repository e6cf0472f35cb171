// trio_cli.cpp -- minimal C++ host driver over include/denovo_kmer.hpp, used by the tests to show
// that a compiled-language host (the stand-in for the reference's Rust CLI) gets the same result
// through the C ABI as the Python binding.
//
//   trio_cli k log2_bits n_hashes seed mode parents.txt child.txt
//   -> one line per child-only k-mer: "hi lo count", unordered; then "STATS n_windows n_valid n_absent n_distinct"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/denovo_kmer.hpp"

static std::vector<std::string> read_lines(const char *path)
{
    std::vector<std::string> out;
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line)) out.push_back(line);
    return out;
}

int main(int argc, char **argv)
{
    if (argc != 8) {
        std::fprintf(stderr, "usage: %s k log2_bits n_hashes seed mode parents.txt child.txt\n", argv[0]);
        return 2;
    }
    try {
        dk_host::Config c;
        c.k = (uint32_t)std::atoi(argv[1]);
        c.filter_log2_bits = (uint32_t)std::atoi(argv[2]);
        c.n_hashes = (uint32_t)std::atoi(argv[3]);
        c.seed = std::strtoull(argv[4], nullptr, 10);
        c.mode = (uint32_t)std::atoi(argv[5]);
        dk_host::Engine eng(c);
        dk_host::KmerSet parents(eng);
        parents.insert_sequences(read_lines(argv[6]));
        dk_host::ReadBatch child(eng, read_lines(argv[7]));
        dk_host::KmerCounter counter(eng);
        dk_host::KmerCounts res = counter.child_only(child, parents);
        for (size_t i = 0; i < res.size(); i++)
            std::printf("%llu %llu %u\n", (unsigned long long)res.hi[i], (unsigned long long)res.lo[i], res.count[i]);
        std::printf("STATS %llu %llu %llu %llu\n", (unsigned long long)res.stats.n_windows,
                    (unsigned long long)res.stats.n_valid, (unsigned long long)res.stats.n_absent,
                    (unsigned long long)res.stats.n_distinct);
    } catch (const dk_host::Error &e) {
        std::fprintf(stderr, "denovo_kmer error %d: %s\n", (int)e.status, e.what());
        return 1;
    }
    return 0;
}
