// ref_harness.cpp -- ORACLE tooling (test infrastructure, NOT product code).
//
// A driver that links the UNMODIFIED reference sources where they lie under
// /root/reference (see oracle/Makefile; nothing is copied into this repo) and
// exposes the reference's own classes on the command line, so golden vectors
// and the "reference" CPU baseline come from the reference implementation
// itself:
//
//   score  <tokens.json> <shard.json> <out.json> [--ids] [--threads N]
//       SolutionFile(path) + DataFile(path).GetEntries() + Tokenize per entry,
//       accumulated as /root/reference/src/main.cpp:31-46 (with the Wait()
//       the reference forgets).  Writes counts / totals (and the id sequence
//       of every document with --ids) as JSON, and the Tokenize wall time.
//   anneal <data_dir> <max_len> <pref> <batch> <passes> <tokens.json>
//       main.cpp:15-26: MetadataFile -> GetTokens -> TokenGenerator ->
//       Generate -> GetSolution -> SolutionFile(solution, path).
//   dag    <data_dir> <max_len> <nsteps> <seed> <out.json>
//       GetTokens on a tiny corpus, then a scripted single-threaded toggle
//       sequence through Token::SimulateStep / Token::ApplyStep<bool>
//       (Token.cpp:83-105), dumping the DAG and every return value.
//   trace  <data_dir> <max_len> <pref> <batch> <seed> <ntasks> <out.json>
//       GetTokens -> TokenGenerator, the calling thread's generator (the
//       thread_local `gen` of TokenGenerator.cpp:52) seeded with <seed>, then
//       <ntasks> TokenGenerator::WorkerTask calls on this thread, the state after
//       each one, and GetSolution at the end.  GetTokens leaves its
//       `.candidates-all-<max_len>.bin` (WriteTokens, Token.cpp:163-193) in <data_dir>.
//
// Only this file is ours; every symbol it calls is the reference's.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <functional>
#include <future>
#include <iostream>
#include <map>
#include <mutex>
#include <queue>
#include <random>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "rapidjson/document.h"
#include "utf8cpp/utf8.h"

// The DAG state (uses, parents, enabled flag, per-token mutex) is private in
// the reference (Token.h:66-77, TokenGenerator.h:16-36).  The harness needs to
// read it to dump golden vectors, so it opens the access specifiers for its
// own translation unit only; layouts and mangled names are unaffected and the
// reference's .cpp files are compiled untouched.
#define private public
#define class struct
#include "files/DataFile.h"
#include "files/MetadataFile.h"
#include "files/SolutionFile.h"
#include "tokenizer/GetTokens.h"
#include "tokenizer/Token.h"
#include "tokenizer/TokenGenerator.h"
#undef class
#undef private

static void json_escape(std::ostream &o, const std::string &s) {
	static const char *hex = "0123456789abcdef";
	o << '"';
	for (unsigned char c : s) {
		if (c == '"') o << "\\\"";
		else if (c == '\\') o << "\\\\";
		else if (c < 0x20) o << "\\u00" << hex[c >> 4] << hex[c & 15];
		else o << (char)c;
	}
	o << '"';
}

static int cmd_score(int argc, char **argv) {
	if (argc < 5) return 2;
	bool want_ids = false;
	int threads = 1;
	for (int i = 5; i < argc; i++) {
		if (!strcmp(argv[i], "--ids")) want_ids = true;
		else if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
	}
	SolutionFile tkn(argv[2]);
	if (!((JsonFile &)tkn).IsValid()) {
		std::cerr << "invalid tokens file " << argv[2] << std::endl;
		return 3;
	}
	DataFile shard(argv[3]);
	if (!shard.IsValid()) {
		std::cerr << "invalid shard " << argv[3] << std::endl;
		return 3;
	}
	const std::vector<DataFile::Entry> entries = shard.GetEntries();
	const size_t D = entries.size();
	std::vector<std::vector<size_t>> all_ids(want_ids ? D : 0);
	std::vector<size_t> doc_tokens(D);

	std::atomic<size_t> init_size = 0; // main.cpp:34
	std::atomic<size_t> comp_size = 0; // main.cpp:35
	std::vector<std::map<size_t, size_t>> counts(threads);
	std::atomic<size_t> next = 0;
	auto t0 = std::chrono::steady_clock::now();
	auto work = [&](int t) {
		for (;;) {
			size_t d = next++;
			if (d >= D) break;
			const std::string &text = entries[d].text;
			init_size += text.size();                       // main.cpp:40
			std::vector<size_t> ids = tkn.Tokenize(text);   // main.cpp:41
			comp_size += ids.size() - 2;
			doc_tokens[d] = ids.size() - 2;
			for (size_t id : ids) counts[t][id]++;
			if (want_ids) all_ids[d] = std::move(ids);
		}
	};
	if (threads <= 1) work(0);
	else {
		std::vector<std::thread> th;
		for (int t = 0; t < threads; t++) th.emplace_back(work, t);
		for (auto &x : th) x.join();
	}
	auto t1 = std::chrono::steady_clock::now();
	double secs = std::chrono::duration<double>(t1 - t0).count();
	std::map<size_t, size_t> total;
	for (auto &m : counts)
		for (auto &kv : m) total[kv.first] += kv.second;

	std::ofstream out(argv[4]);
	out << "{\"docs\": " << D << ", \"total_bytes\": " << init_size
	    << ", \"total_tokens\": " << comp_size << ", \"tokenize_seconds\": " << secs
	    << ", \"threads\": " << threads << ",\n \"counts\": {";
	bool first = true;
	for (auto &kv : total) {
		if (!first) out << ", ";
		first = false;
		// (size_t)-1 is the reference's UNKNOWN id (SolutionFile.cpp:76)
		if (kv.first == (size_t)-1) out << "\"-1\": " << kv.second;
		else out << '"' << kv.first << "\": " << kv.second;
	}
	out << "},\n \"doc_tokens\": [";
	for (size_t d = 0; d < D; d++) out << (d ? ", " : "") << doc_tokens[d];
	out << "]";
	if (want_ids) {
		out << ",\n \"ids\": [";
		for (size_t d = 0; d < D; d++) {
			out << (d ? ",\n  [" : "\n  [");
			for (size_t i = 0; i < all_ids[d].size(); i++) {
				if (i) out << ", ";
				if (all_ids[d][i] == (size_t)-1) out << -1;
				else out << all_ids[d][i];
			}
			out << "]";
		}
		out << "]";
	}
	out << "}\n";
	return 0;
}

// text: SolutionFile::Detokenize / Prettify (SolutionFile.cpp:102-118) of every document's ids,
// hex-encoded (the strings may hold any byte).
static int cmd_text(int argc, char **argv) {
	if (argc < 5) return 2;
	SolutionFile tkn(argv[2]);
	DataFile shard(argv[3]);
	if (!((JsonFile &)tkn).IsValid() || !shard.IsValid()) return 3;
	auto hex = [](const std::string &s) {
		static const char *d = "0123456789abcdef";
		std::string h;
		for (unsigned char c : s) h += d[c >> 4], h += d[c & 15];
		return h;
	};
	std::ofstream out(argv[4]);
	out << "{\"detokenize_hex\": [";
	std::vector<std::string> pretty;
	bool first = true;
	for (const DataFile::Entry &e : shard.GetEntries()) {
		const std::vector<size_t> ids = tkn.Tokenize(e.text);
		out << (first ? "" : ", ") << '"' << hex(tkn.Detokenize(ids)) << '"';
		pretty.push_back(hex(tkn.Prettify(ids)));
		first = false;
	}
	out << "],\n \"prettify_hex\": [";
	for (size_t i = 0; i < pretty.size(); i++) out << (i ? ", " : "") << '"' << pretty[i] << '"';
	out << "]}\n";
	return 0;
}

// Generate() aborts as soon as stdin is readable (TokenGenerator.cpp:186);
// park fd 0 on a pipe whose write end stays open so it never is.
static void silence_stdin() {
	int fds[2];
	if (pipe(fds) == 0) dup2(fds[0], 0);
}

static int cmd_anneal(int argc, char **argv) {
	if (argc < 8) return 2;
	silence_stdin();
	const std::string dir = argv[2];
	const int max_len = atoi(argv[3]);
	const size_t pref = strtoull(argv[4], nullptr, 10);
	const size_t batch = strtoull(argv[5], nullptr, 10);
	const size_t passes = strtoull(argv[6], nullptr, 10);
	MetadataFile metadata(dir + "/.metadata.json", true);             // main.cpp:15
	std::vector<std::string> solution;
	{
		std::vector<annealing::Token> tokens =
		    annealing::GetTokens(metadata, (uint8_t)max_len, -1, true); // main.cpp:20
		annealing::TokenGenerator generator(std::move(tokens), pref, batch); // :21
		auto t0 = std::chrono::steady_clock::now();
		generator.Generate(passes);                                   // :22
		const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		// moves = gen_cnt_ increments / wall time
		std::cerr << "{\"generate_seconds\": " << secs << ", \"gen_cnt\": " << (size_t)generator.gen_cnt_
		          << ", \"tot_cand\": " << generator.tot_cand_ << ", \"enabled_cnt\": " << (size_t)generator.enabled_cnt_
		          << ", \"threads\": " << std::min((uint64_t)std::thread::hardware_concurrency(), generator.tot_cand_ / batch)
		          << "}" << std::endl;
		solution = generator.GetSolution();                           // :24
	}
	SolutionFile tkn(solution, argv[7]);                              // :26
	return 0;
}

static int cmd_dag(int argc, char **argv) {
	if (argc < 7) return 2;
	silence_stdin();
	const std::string dir = argv[2];
	const int max_len = atoi(argv[3]);
	const size_t nsteps = strtoull(argv[4], nullptr, 10);
	const uint64_t seed = strtoull(argv[5], nullptr, 10);
	MetadataFile metadata(dir + "/.metadata.json", true);
	std::vector<annealing::Token> tokens = annealing::GetTokens(metadata, (uint8_t)max_len, -1, true);
	const size_t n = tokens.size();
	// TokenGenerator's ctor wires the striped mutexes and enables the
	// single-character roots (TokenGenerator.cpp:30-45); the vector move keeps
	// the Token addresses, so parent pointers stay valid.
	annealing::TokenGenerator gen(std::move(tokens), 30000, 30);
	std::vector<annealing::Token> &tk = gen.tokens_;
	const annealing::Token *base = tk.data();

	std::ofstream out(argv[6]);
	out << "{\"max_len\": " << max_len << ", \"tot_cand\": " << gen.tot_cand_ << ",\n \"tokens\": [";
	for (size_t i = 0; i < n; i++) {
		const annealing::Token &t = tk[i];
		out << (i ? ",\n  " : "\n  ") << "{\"name\": ";
		json_escape(out, t.GetName());
		out << ", \"chr\": " << (uint32_t)t.chr_ << ", \"size\": " << t.size()
		    << ", \"uses\": " << (uint64_t)t.l_branch_.uses << ", \"l\": "
		    << (t.l_branch_.parent ? (long long)(t.l_branch_.parent - base) : -1) << ", \"r\": "
		    << (t.r_branch_.parent ? (long long)(t.r_branch_.parent - base) : -1)
		    << ", \"enabled\": " << (t.enabled_ ? 1 : 0) << "}";
	}
	out << "],\n \"steps\": [";
	uint64_t s = seed * 6364136223846793005ull + 1442695040888963407ull;
	out.precision(17);
	size_t done = 0;
	for (size_t k = 0; k < nsteps && gen.tot_cand_ > 0; k++) {
		s = s * 6364136223846793005ull + 1442695040888963407ull;
		size_t i = (size_t)((s >> 33) % n);
		annealing::Token &t = tk[i];
		if (t.size() == 1) continue; // roots are never toggled (TokenGenerator.cpp:35-39)
		const bool enable = !t.enabled_;
		const double sim = t.SimulateStep();
		const double app = enable ? t.ApplyStep<true>() : t.ApplyStep<false>();
		out << (done++ ? ",\n  " : "\n  ") << "{\"i\": " << i << ", \"enable\": " << (enable ? 1 : 0)
		    << ", \"sim\": " << sim << ", \"apply\": " << app << "}";
	}
	out << "],\n \"final\": [";
	for (size_t i = 0; i < n; i++) {
		const annealing::Token &t = tk[i];
		out << (i ? ", " : "") << "[" << (uint64_t)t.l_branch_.uses << ", " << (uint64_t)t.r_branch_.uses
		    << ", " << (t.enabled_ ? 1 : 0) << "]";
	}
	out << "]}\n";
	return 0;
}

// cands <dir> <max_len> <out.json>: the candidate list GetTokens builds from the shards of
// <dir> (GetTokens.cpp:92-115), names and uses (= Trie freq, Trie.cpp:82) in its own order.
static int cmd_cands(int argc, char **argv) {
	if (argc < 5) return 2;
	const std::string dir = argv[2];
	const int max_len = atoi(argv[3]);
	MetadataFile metadata(dir + "/.metadata.json", true);
	std::vector<annealing::Token> tokens = annealing::GetTokens(metadata, (uint8_t)max_len, -1, true);
	std::ofstream out(argv[4]);
	out << "{\"max_len\": " << max_len << ",\n \"tokens\": [";
	for (size_t i = 0; i < tokens.size(); i++) {
		out << (i ? ",\n  [" : "\n  [");
		json_escape(out, tokens[i].GetName());
		out << ", " << (uint64_t)tokens[i].l_branch_.uses << "]";
	}
	out << "]}\n";
	return 0;
}

// TokenGenerator.cpp:51-53 defines these at namespace scope with external linkage
extern thread_local std::mt19937 gen;

static int cmd_trace(int argc, char **argv) {
	if (argc < 9) return 2;
	silence_stdin();
	const std::string dir = argv[2];
	const int max_len = atoi(argv[3]);
	const size_t pref = strtoull(argv[4], nullptr, 10);
	const size_t batch = strtoull(argv[5], nullptr, 10);
	const uint32_t seed = (uint32_t)strtoull(argv[6], nullptr, 10);
	const size_t ntasks = strtoull(argv[7], nullptr, 10);
	MetadataFile metadata(dir + "/.metadata.json", true);
	std::vector<annealing::Token> tokens = annealing::GetTokens(metadata, (uint8_t)max_len, -1, true);
	annealing::TokenGenerator g(std::move(tokens), pref, batch);
	gen.seed(seed);
	std::ofstream out(argv[8]);
	out.precision(17);
	auto dump = [&](bool first) {
		double beta, sigma;
		g.score_dist_.GetParams(&beta, &sigma);
		out << (first ? "\n  " : ",\n  ") << "{\"gen_cnt\": " << (size_t)g.gen_cnt_ << ", \"enabled_cnt\": " << (size_t)g.enabled_cnt_
		    << ", \"raw_score\": " << (double)g.raw_score_ << ", \"temp\": " << (double)g.temp_
		    << ", \"moment1\": " << g.score_dist_.moment1_ << ", \"moment2\": " << g.score_dist_.moment2_
		    << ", \"beta\": " << beta << ", \"sigma\": " << sigma << "}";
	};
	out << "{\"max_len\": " << max_len << ", \"pref\": " << pref << ", \"batch\": " << batch << ", \"seed\": " << seed
	    << ", \"tot_cand\": " << g.tot_cand_ << ", \"n_tokens\": " << g.tokens_.size() << ",\n \"states\": [";
	for (size_t k = 0; k < ntasks; k++) {
		g.WorkerTask();
		dump(k == 0);
	}
	out << "],\n \"solution\": [";
	const std::vector<std::string> sol = g.GetSolution();
	for (size_t i = 0; i < sol.size(); i++) {
		out << (i ? ", " : "");
		json_escape(out, sol[i]);
	}
	out << "]}\n";
	return 0;
}

// load <shard.json>: DataFile(path) + GetEntries() (JsonFile.cpp:14-29, DataFile.cpp:3-31), timed
static int cmd_load(int argc, char **argv) {
	if (argc < 3) return 2;
	auto t0 = std::chrono::steady_clock::now();
	DataFile shard(argv[2]);
	size_t docs = 0, bytes = 0;
	if (shard.IsValid())
		for (const DataFile::Entry &e : shard.GetEntries()) docs++, bytes += e.text.size();
	const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	std::cout << "{\"valid\": " << (shard.IsValid() ? 1 : 0) << ", \"docs\": " << docs << ", \"text_bytes\": " << bytes
	          << ", \"seconds\": " << secs << "}" << std::endl;
	return 0;
}

int main(int argc, char **argv) {
	if (argc < 2) {
		std::cerr << "usage: ref_harness score|anneal|dag|cands|trace|load ..." << std::endl;
		return 2;
	}
	const std::string cmd = argv[1];
	if (cmd == "score") return cmd_score(argc, argv);
	if (cmd == "anneal") return cmd_anneal(argc, argv);
	if (cmd == "dag") return cmd_dag(argc, argv);
	if (cmd == "cands") return cmd_cands(argc, argv);
	if (cmd == "trace") return cmd_trace(argc, argv);
	if (cmd == "load") return cmd_load(argc, argv);
	if (cmd == "text") return cmd_text(argc, argv);
	std::cerr << "unknown command " << cmd << std::endl;
	return 2;
}
