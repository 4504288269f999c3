// Drop-in RUFUS.Build (legacy set difference on sorted TEXT tables, src/RUFUS.Build.cpp:224-283):
//   RUFUS.Build -c ParentTable... -s SubjectTable [-o OUT] -hs K -mS minSubject -mC maxControl [-max n] [-t n] [-d c]
// For every subject line `kmer<TAB>count` with mS <= count <= max, each parent cursor advances while
// its k-mer is lexicographically smaller; parent counts of equal k-mers are summed; the line
// `HashToLong(kmer)<TAB>parentDepth<TAB>count<TAB>kmer` is written when parentDepth <= mC.
// A streaming merge-join of text: host work by nature (the GPU path of the same set difference is
// `jellyfish merge` + CheckJellyHashList, which runRufus.sh actually calls).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

namespace {
// Util::Split (src/Util.cpp:24-33)
std::vector<std::string> split(const std::string& s, char d) {
  std::vector<std::string> t;
  size_t i = 0;
  while (i < s.size()) {
    const size_t j = s.find(d, i);
    if (j == std::string::npos) { t.push_back(s.substr(i)); break; }
    t.push_back(s.substr(i, j - i));
    i = j + 1;
  }
  return t;
}
// Util::HashToLong (src/Util.cpp:51-84): base i at bits (2i, 2i+1), A 00, C hi, G lo, T both
unsigned long hash_to_long(const std::string& s) {
  unsigned long v = 0;
  for (size_t i = 0; i < s.size() && i < 32; ++i) {
    unsigned long lo = 0, hi = 0;
    if (s[i] == 'C') hi = 1;
    else if (s[i] == 'G') lo = 1;
    else if (s[i] == 'T') lo = hi = 1;
    else if (s[i] != 'A') std::cout << "ERROR, invalid character - " << s[i] << std::endl;
    v |= lo << (2 * i) | hi << (2 * i + 1);
  }
  return v;
}
const std::string& field(const std::vector<std::string>& v, size_t i) {
  static const std::string empty;
  return i < v.size() ? v[i] : empty;
}
}  // namespace

int main(int argc, char** argv) {
  std::vector<std::string> parents;
  std::string subject, out;
  int k = -1;
  double min_subject = -1.0, max_control = -1.0;
  long max_cov = 100000000;
  char delim = '\t';
  for (int i = 1; i < argc; ++i) {
    const std::string p = argv[i];
    auto need = [&](int n) { return i + n < argc; };
    if (p == "-h") { std::cout << "RUFUS.Build -c F... -s F -o OUT -hs K -mS n -mC n [-max n] [-t n] [-d c]\n"; return 0; }
    else if (p == "-c" && need(1)) parents.push_back(argv[++i]);
    else if (p == "-s" && need(1)) subject = argv[++i];
    else if (p == "-o" && need(1)) out = argv[++i];
    else if (p == "-hs" && need(1)) k = atoi(argv[++i]);
    else if (p == "-mS" && need(1)) min_subject = atof(argv[++i]);
    else if (p == "-mC" && need(1)) max_control = atof(argv[++i]);
    else if (p == "-max" && need(1)) max_cov = atoi(argv[++i]);
    else if (p == "-t" && need(1)) ++i;
    else if (p == "-d" && need(1)) delim = argv[++i][0];
    else { std::cout << "ERROR: unkown command line paramater -" << argv[i] << "-" << std::endl; return 0; }
  }
  if (parents.empty()) { std::cout << "Error in control file inputs" << std::endl; return 0; }
  if (subject.empty()) { std::cout << "Error subject file required" << std::endl; return 0; }
  if (out.empty()) out = subject;
  if (k == -1) { std::cout << "Error Hash Size required" << std::endl; return 0; }
  if (min_subject < 0) { std::cout << "Error Minimum coverage in the subject required" << std::endl; return 0; }
  if (max_control < -1) { std::cout << "Error Maximum Coverage in the control must be set" << std::endl; return 0; }

  const std::string sentinel = std::string((size_t)k, 'T') + "0";  // what an exhausted parent reads as (:150-156)
  std::vector<std::ifstream> pf(parents.size());
  for (size_t i = 0; i < parents.size(); ++i) {
    pf[i].open(parents[i].c_str());
    if (!pf[i].is_open()) { std::cout << "Error, ParentHashFile could not be opened" << std::endl << parents[i] << std::endl; return 0; }
  }
  std::ifstream sf(subject.c_str());
  if (!sf.is_open()) { std::cout << "Error, MutHashFile could not be opened"; return 0; }
  std::ofstream of(out.c_str());
  std::ofstream log((out + ".Buld.Log").c_str());
  if (!of.is_open() || !log.is_open()) { std::cout << "Error MutHashTableFilter file couldnt be opened - " << out << std::endl; return 0; }

  std::vector<std::string> cur(parents.size());
  for (size_t i = 0; i < parents.size(); ++i) std::getline(pf[i], cur[i]);
  std::string line;
  while (std::getline(sf, line)) {
    const std::vector<std::string> m = split(line, '\t');
    const int a = atoi(field(m, 1).c_str());
    if (!(a >= min_subject && a <= max_cov)) continue;
    int depth = 0;
    for (size_t i = 0; i < parents.size(); ++i) {
      std::vector<std::string> p = split(cur[i], delim);
      while (field(p, 0) < field(m, 0)) {
        const std::string last = cur[i];
        std::getline(pf[i], cur[i]);
        if (cur[i].empty()) {
          std::cout << "Parent " << parents[i] << " died at " << last << std::endl;
          cur[i] = sentinel;
        }
        p = split(cur[i], delim);
      }
      if (field(p, 0) == field(m, 0)) depth += atoi(field(p, 1).c_str());
    }
    if (depth <= max_control) of << hash_to_long(field(m, 0)) << "\t" << depth << "\t" << a << "\t" << field(m, 0) << std::endl;
  }
  std::cout << "\nDone reading Parent File\n\nreally done\n";
  return 0;
}
