// my_grep -- the program of the reference's README (README.md:31-41: "building a (basic) grep-like executable using
// x-search"), compiled against include/xsearch/xsearch.h of this repository with nothing changed in it: the one call
// xs::extern_search<xs::lines>(pattern, file, false, 1) and the live range-for over getResult().  It is what the
// reference's only published measurement times against GNU grep (README.md:44-62); bench.py's `cli` block times this
// binary the same way.
#include <xsearch/xsearch.h>
#include <iostream>

int main(int argc, char** argv) {
  auto searcher = xs::extern_search<xs::lines>(argv[1], argv[2], false, 1);
  for (auto const& line : *searcher->getResult()) {
    std::cout << line << '\n';
  }
}
