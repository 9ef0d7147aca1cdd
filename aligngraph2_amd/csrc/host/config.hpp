// config.txt of the -p directory: blocks of
//   reference name / read file / read->contig ALN / read->ref ALN / (contig name, 1|0)* / blank line
// Restates loadFromConfig (reference PAGraph/src/main/pagraph.cpp:29-49).
#pragma once
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "raw_input.hpp"

namespace pagh {

inline std::vector<BlockConfig> loadConfig(const std::string &path) {
    std::vector<BlockConfig> blocks;
    std::ifstream in(path);
    std::string line;
    while (std::getline(in, line)) {
        BlockConfig b;
        b.ref = line;
        std::getline(in, b.readPath);
        std::getline(in, b.ctgAlnPath);
        std::getline(in, b.refAlnPath);
        while (std::getline(in, line) && !line.empty()) {
            b.contigs.emplace_back(line, false);
            std::getline(in, line);
            std::stringstream(line) >> b.contigs.back().second;
        }
        blocks.push_back(b);
    }
    return blocks;
}

}  // namespace pagh
