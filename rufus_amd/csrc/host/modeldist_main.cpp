// ModelDist HISTO K ReadLength Threads -- drop-in for the reference's coverage model fit (src/ModelDist.cpp,
// called at runRufus.sh:849 on every sample's `jellyfish histo` table; :862-868 read lines 2 and 4 of
// HISTO.7.7.model as MutantMinCov and MutantSC, Overlap.shorter.sh:346 hands HISTO.7.7.dist to RUFUS.interpret).
//
// The fit itself -- ~70 dependent steps of 11 candidate models, 4*10^7 normal densities per step, 40 s on the
// reference's 11 OpenMP threads -- runs behind rfx_model_residuals / rfx_model_tables (csrc/rfx_model.hip).  This
// file parses the table, fits the 1/x^p error curve (a few thousand libm calls), drives the search and writes the
// three files and the stdout log in the reference's layout.  No CPU fallback: without a gfx950 device it stops.
//
// Kept from the reference because they shape the numbers downstream tools read: the histogram vector starts at
// the first non-empty row, so its index is not the multiplicity when row 1 is empty (:437-452); pi = 3.14159; the
// skew search never runs (:605 tests `SKhigh < 1e-50`); `float` accumulators in the error fit and the cutoff;
// a `long` accumulator on line 5 of the model file (:885-893); main()'s `prob` is one row behind `dist` (:684).
// Not emulated (undefined behaviour there): the 11 values written into `double values[9]` (:546), the cell one
// past the last row of `prob` (:805, taken as 0), `ErrorDist[n]` on the last line of the .prob file (:955, 0).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "rfx_cli.hpp"

namespace {

using rfxcli::die;

std::vector<std::string> tab_fields(const std::string& line) {
  std::vector<std::string> out;
  std::stringstream ss(line);
  std::string tok;
  while (std::getline(ss, tok, '\t')) out.push_back(tok);
  return out;
}

struct Histogram {
  std::vector<int64_t> at;  // [0] = 0, [1] = first non-empty row of the file, ...
  double peak = 1, peak_value = -1, sum = 0;  // SC, SCvalue, HistoSum (:406-411)
  long total = 0, total_kmers = 0;
  int inflection = -1;
  long last_index = 1;
};

// :428-478, with its log lines
int read_histogram(std::istream& in, Histogram& h) {
  std::string line;
  std::getline(in, line);
  std::vector<std::string> f = tab_fields(line);
  if (f.size() < 2) die("ModelDist: the histogram is not a two-column, tab-separated table (runRufus.sh:830 makes it one)");
  std::cout << "first line = " << f[0] << " - " << f[1] << std::endl;
  int burnt = 0;
  while (atoi(f[1].c_str()) == 0 || atoi(f[0].c_str()) == 0) {
    std::cout << "getting another " << std::endl;
    if (!std::getline(in, line)) line.clear();
    f = tab_fields(line);
    if (f.size() < 2) f.resize(2);
    std::cout << "got " << f[0] << " - " << f[1] << std::endl;
    if (++burnt > 10) {
      std::cout << "ERROR there are no kmers in this file" << std::endl;
      return 1;
    }
  }
  std::cout << "going with " << f[0] << " - " << f[1] << std::endl;
  double value = (double)atol(f[1].c_str());
  h.at = {0, (int64_t)value};
  long prev = (long)value;
  bool rising = false;
  long i = 1;
  while (std::getline(in, line)) {
    ++i;
    f = tab_fields(line);
    if (f.size() < 2) die("ModelDist: row " + std::to_string(i) + " of the histogram has no count column");
    value = (double)atol(f[1].c_str());
    h.at.push_back((int64_t)value);
    h.total += value;  // (long += double, as there)
    h.total_kmers += value * atoi(f[0].c_str());
    h.sum += value;
    if (value - prev > 0 && !rising) {
      h.inflection = (int)(i - 1);
      rising = true;
    }
    if (rising && h.peak_value < value) {
      h.peak_value = value;
      h.peak = (double)i;
    }
    prev = (long)h.at[(size_t)i];
  }
  h.last_index = i;
  return 0;
}

// FitErrorModel (:339-365): the power p of Error[1] / i^p closest (in log space) to the rows below the inflection
float fit_error_curve(std::vector<double>& err, double& total, int upto) {
  auto ssq_for = [&](double p) {
    double s = 0;
    for (int i = 1; i < upto; ++i) s += pow(log(err[(size_t)i]) - log((1 / (pow(i, p))) * err[1]), 2);
    return s;
  };
  double best_ssq = ssq_for(100);
  float best_p = 0;
  for (float p = 7; p > .1; p += -.001) {
    const double s = ssq_for(p);
    if (s < best_ssq) {
      best_ssq = s;
      best_p = p;
    }
  }
  total = 0;
  for (size_t i = 1; i < err.size(); ++i) {
    err[i] = (1 / (pow((double)i, best_p))) * err[1];
    total += err[i];
  }
  std::cout << "best error is 1/x^" << best_p << std::endl;
  return best_p;
}

struct Search {
  rfx_ctx* ctx;
  const std::vector<int64_t>& histo;  // error curve subtracted (histo2, :511-516)
  int inflection;

  // 11 candidates on [lo, hi] -> their residuals, one device pass
  template <class Make>
  int best_of(double lo, double hi, bool log_resid, Make&& make) const {
    rfx_model_params cand[11];
    double resid[11];
    for (int x = 0; x <= 10; ++x) cand[x] = make(lo + (((hi - lo) / 10) * x));
    const int rc = rfx_model_residuals(ctx, histo.data(), (uint32_t)histo.size(), cand, 11, log_resid ? 1 : 0, inflection, 5,
                                       resid);
    if (rc == RFX_E_INVAL || rc == RFX_E_RANGE)
      die("ModelDist: a candidate model (SC " + std::to_string(cand[0].sc) + " .. " + std::to_string(cand[10].sc) +
          ") does not fit a histogram of " + std::to_string(histo.size()) +
          " rows (the reference reads outside its tables here)");
    if (rc != RFX_OK) die(std::string("ModelDist: rfx_model_residuals failed: ") + rfx_last_error());
    int at = 0;
    for (int x = 1; x <= 10; ++x)
      if (resid[x] < resid[at]) at = x;
    return at;
  }

  // the interval refinement every parameter goes through (:540-570 and its four repeats): keep the two cells
  // around the best of 11 points until lo / hi reaches `ratio`
  template <class Make>
  double refine(double lo, double hi, double ratio, double hi_floor, double lo_clamp, bool log_resid, double& best,
                Make&& make) const {
    double steps = 0;
    while (lo / hi < ratio && hi > hi_floor) {
      ++steps;
      const int at = best_of(lo, hi, log_resid, make);
      const double below = lo + ((hi - lo) / 10) * (at - 1);
      lo = below >= lo_clamp ? below : lo_clamp;
      hi = lo + ((hi - lo) / 10) * (at + 1);
      best = lo + ((hi - lo) / 10) * at;
    }
    return steps;
  }
};

// `ostream << double` for the big tables, without the stream machinery (6.9e6 cells, most of them 0)
struct Cells {
  std::string buf;
  void num(double v) {
    if (v == 0 && !std::signbit(v)) {
      buf.push_back('0');
      return;
    }
    char tmp[40];
    buf.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "%g", v));
  }
  void flush_to(std::ofstream& f) {
    f.write(buf.data(), (std::streamsize)buf.size());
    buf.clear();
  }
};

}  // namespace

int main(int argc, char** argv) {
  std::cout << "Call is histoFile HS ReadLength Threads" << std::endl;
  if (argc < 5) {
    std::cerr << "usage: ModelDist HISTO K ReadLength Threads" << std::endl;
    return 1;
  }
  std::ifstream in(argv[1]);
  if (in.is_open()) {
    std::cout << "Parent File open - " << argv[1] << std::endl;
  } else {
    std::cout << "Error, HistoFile could not be opened";
    return 0;  // (:378)
  }
  const std::string stem = argv[1];
  std::ofstream model_file(stem + ".7.7.model"), dist_file(stem + ".7.7.dist");
  if (!model_file.is_open() || !dist_file.is_open()) {
    std::cout << "Error, Model file could not be opened";
    return 0;
  }
  const int k = atoi(argv[2]), read_length = atoi(argv[3]);

  Histogram h;
  if (read_histogram(in, h)) return 1;
  const std::vector<int64_t>& histo = h.at;
  const size_t n = histo.size();
  if (n < 12) die("ModelDist: a histogram of " + std::to_string(n) + " rows is too short to fit");
  const long reads = read_length - k + 1 != 0 ? h.total_kmers / (read_length - k + 1) : 0;
  std::cout << "Number of reads = " << (int)reads << std::endl;
  for (int i = 0; i < 10; ++i) std::cout << "I = " << i << " \t " << histo[(size_t)i] << std::endl;
  std::cout << "SC = " << h.peak << " vlaue = " << h.peak_value << std::endl;
  const int raw_sc = (int)h.peak;
  if (h.inflection < 1) die("ModelDist: the histogram never rises: no inflection point, nothing to fit");

  // one standard deviation = where the peak has fallen to e^-1/2 of its height (:488-498)
  const double sd_level = h.peak_value * exp(-.5);
  long sd_at = (long)h.peak;
  while (sd_at < (long)n && !(histo[(size_t)sd_at] - sd_level < 0)) ++sd_at;
  double stdev = sd_at - h.peak;
  std::cout << "stdi = " << sd_at << " stdev = " << stdev << std::endl;

  std::vector<double> error_curve(histo.begin(), histo.end());
  double error_total = 0;
  fit_error_curve(error_curve, error_total, h.inflection);
  std::vector<double> error_dist(n);
  std::vector<int64_t> cleaned(histo);
  for (size_t i = 0; i < n; ++i) {
    error_dist[i] = error_curve[i] / error_total;
    cleaned[i] = histo[i] - error_curve[i] > 0 ? (int64_t)(histo[i] - error_curve[i]) : 0;
  }

  rfx_ctx* ctx = rfxcli::open_ctx();
  rfxcli::trace("device open");
  double sc = h.peak, factor = 1, skew = 0, power = 1;
  double best_sd = stdev, best_f = factor, best_sc = sc, best_sk = skew, best_p = power;
  const Search search{ctx, cleaned, h.inflection};
  for (int pass = 0; pass <= 2; ++pass) {
    std::cout << "On " << pass + 1 << " pass" << std::endl;
    double steps = search.refine(1, 20, .999, 1e-10, 0, true, best_f,
                                 [&](double v) { return rfx_model_params{best_sc, best_sd, v, best_sk, best_p}; });
    std::cout << "\t best Factor = " << best_f << " steps = " << steps << std::endl;
    steps = search.refine(sc * .9, sc * 1.1, .999, 1e-50, 0, false, best_sc,
                          [&](double v) { return rfx_model_params{v, best_sd, best_f, best_sk, best_p}; });
    std::cout << "\t\tbestSC = " << best_sc << " steps = " << steps << std::endl;
    steps = search.refine(stdev * .9, stdev * 1.1, .99, 1e-50, 0, false, best_sd,
                          [&](double v) { return rfx_model_params{best_sc, v, best_f, best_sk, best_p}; });
    std::cout << "\t\tbest StdDev = " << best_sd << " steps = " << steps << std::endl;
    std::cout << "\t\tbest skew factor = " << best_sk << " steps = " << 0 << std::endl;  // (:605: never entered)
    steps = search.refine(1, 2, .999, 1e-50, 1, true, best_p,
                          [&](double v) { return rfx_model_params{best_sc, best_sd, best_f, best_sk, v}; });
    std::cout << "\t\tbest Power factor = " << best_p << " steps = " << steps << std::endl;
    stdev = best_sd, factor = best_f, sc = best_sc, skew = best_sk, power = best_p;
  }
  rfxcli::trace("search done");
  std::cout << "Best Model is SC = " << sc << " StdDev = " << stdev << " F = " << factor << " skew = " << skew
            << " bestP = " << power << std::endl;

  // ---- tables of the chosen model (:702-826) ----
  const rfx_model_params chosen{sc, stdev, factor, skew, power};
  uint32_t n_cols = 0;
  int rc = rfx_model_tables(ctx, (uint32_t)n, &chosen, &n_cols, nullptr, 0, nullptr);
  if (rc != RFX_E_RANGE) die(std::string("ModelDist: rfx_model_tables: ") + (rc == RFX_E_INVAL ? "bad model" : rfx_last_error()));
  const size_t width = (size_t)n_cols + 1;  // column 0 (zero), half-copy, 1x .. (n_cols-1)x
  const int copies = (int)n_cols - 1;       // columns the reference normalises and sums (j < n / SC)
  if (copies < 9) die("ModelDist: fewer than 9 copy-number columns fit the histogram (the model file prints 1x..9x)");
  if (!(sc * 5 <= (double)n)) die("ModelDist: 5 x SC lies beyond the histogram");
  std::vector<double> dist(n * width), rowtot(n);
  rc = rfx_model_tables(ctx, (uint32_t)n, &chosen, &n_cols, dist.data(), dist.size(), rowtot.data());
  if (rc != RFX_OK) die(std::string("ModelDist: rfx_model_tables failed: ") + rfx_last_error());
  rfxcli::trace("tables fetched");
  auto D = [&](size_t row, size_t c) -> double { return dist[row * width + c]; };

  std::vector<double> weight(width + 1, 0.0);  // RC: k-mers per copy number
  {
    const size_t s = (size_t)sc, half = (size_t)(sc / 2);
    const double at_peak = histo[s] / D(s, 2);
    const double het = (histo[half] - (D(half, 2) * at_peak)) / D(half, 1);
    weight[1] = het > 0 ? het : 0;
    weight[2] = histo[s] / D(s, 2);
    for (long a = 2; a <= copies; ++a) {
      const size_t r = (size_t)(sc * a);
      const double share = a < copies ? D(r - 1, (size_t)a + 1) / rowtot[r - 1] : 0.0;  // prob is one row behind; last: see top
      weight[(size_t)a + 1] = ((double)histo[r] / D(r, (size_t)a + 1) * share);
    }
  }
  auto M = [&](size_t row, size_t c) -> double { return D(row, c) * weight[c]; };  // model[row][c], c <= copies
  std::vector<double> model_sum(n, 0.0);
  for (size_t i = 0; i < n; ++i) {
    double s = 0;
    for (int c = 1; c <= copies; ++c) s += M(i, (size_t)c);
    model_sum[i] = s;
  }
  double genome = 0;
  for (size_t i = 1; i <= (size_t)copies + 1; ++i) genome += weight[i] * i;
  std::cout << "GenomeSize = " << genome << std::endl;

  int cutoff = -1;  // first row more likely real than error (:836-853)
  for (size_t row = 1; row < n; ++row) {
    float real = 0;
    for (size_t c = 1; c <= n_cols; ++c) real += D(row, c);
    std::cout << "prob not error = " << real / (real + error_dist[row]) << std::endl;
    if (real / (real + error_dist[row]) > 0.5) {
      cutoff = (int)row;
      std::cout << "this one" << std::endl;
      break;
    }
  }
  std::ofstream prob_file(stem + ".7.7.prob");
  if (!prob_file.is_open()) {
    std::cout << "Error, Prob file could not be opened";
    return 0;
  }
  std::cout << "here1" << std::endl;
  for (std::ofstream* f : {&model_file, &dist_file, &prob_file})
    *f << 3 << std::endl << cutoff << std::endl << h.sum << std::endl << raw_sc << std::endl;
  std::cout << "here" << std::endl;

  for (int c = 1; c <= copies; ++c) {
    long share = 0;
    for (size_t row = 1; row < n; ++row) share += M(row, (size_t)c);
    model_file << ((double)share) / ((double)h.sum) << '\t';
  }
  model_file << std::endl;
  model_file << "K\tRawCount\tErrorModel\tContSubtract\tModelSum\t1x\t2x\t3x\t4x\t5x\t6x\t7x\t8x\t9x" << std::endl;
  model_file << 0 << '\t' << 0 << '\t' << 0 << '\t' << 0 << '\t' << 0;
  for (int c = 1; c < 10; ++c) model_file << '\t' << M(0, (size_t)c);
  model_file << std::endl;
  for (long row = 1; row < sc * 5; ++row) {
    model_file << row << '\t' << histo[(size_t)row] << '\t' << error_curve[(size_t)row] << '\t' << 0 << '\t'
               << model_sum[(size_t)row];
    for (int c = 1; c < 10; ++c) model_file << '\t' << M((size_t)row, (size_t)c);
    model_file << std::endl;
  }
  model_file.close();

  dist_file << error_total << '\t' << 0 << '\t';
  for (size_t i = 1; i <= (size_t)copies + 1; ++i) dist_file << weight[i] << '\t';
  dist_file << std::endl << sc << std::endl;
  Cells cells;
  auto dist_row = [&](size_t row) {
    for (size_t c = 1; c <= n_cols; ++c) {
      cells.buf.push_back('\t');
      cells.num(D(row, c));
    }
  };
  cells.buf = "0\t0\t0";
  dist_row(0);
  cells.buf.push_back('\n');
  for (size_t row = 1; row < n; ++row) {
    cells.buf += std::to_string(row);
    cells.buf.push_back('\t');
    cells.num(error_dist[row]);
    cells.buf += "\t0";
    dist_row(row);
    cells.buf.push_back('\n');
    if (cells.buf.size() > (1u << 20)) cells.flush_to(dist_file);
  }
  dist_row(0);  // (:944-946: row 0 once more, no line end)
  cells.flush_to(dist_file);
  dist_file.close();

  prob_file << std::endl << std::endl;
  for (size_t line = 1; line <= n; ++line) {  // line `line` holds row line-1 of the table (:684)
    cells.buf += std::to_string(line);
    cells.buf.push_back('\t');
    cells.num(line < n ? error_dist[line] : 0.0);
    cells.buf += "\t0";
    for (int c = 1; c <= copies; ++c) {
      cells.buf.push_back('\t');
      cells.num(D(line - 1, (size_t)c) / rowtot[line - 1]);
    }
    cells.buf.push_back('\n');
    if (cells.buf.size() > (1u << 20)) cells.flush_to(prob_file);
  }
  cells.flush_to(prob_file);
  prob_file.close();
  rfxcli::trace("files written");

  std::cout << "GenomeSize = " << genome << std::endl;
  std::cout << "Inflection point = " << h.inflection << std::endl;
  std::cout << "Recomended RUFUS cutoff = " << sc - (5 * stdev) << std::endl;
  std::cout << "-1std = " << sc - (1 * stdev) << "\t-2std = " << sc - (2 * stdev) << "\t-3std = " << sc - (3 * stdev)
            << "\t-4std = " << sc - (4 * stdev) << std::endl;
  rfx_close(ctx);
  return 0;
}
