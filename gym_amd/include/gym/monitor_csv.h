// monitor_csv.h — Monitor's CSV log for the batched env (wrappers/monitor.h:20-35,44-50,93-94).
//
// In the reference every env owns a Monitor and, when a file name is given, its own `<name>.csv`:
//     t_start: <clock ticks>\nid: <id>\nt,r,l<first row>\n<row>\n...        row = "<t>,<r>,<l>", floats at ostream default precision
// (the header line `t,r,l` is not followed by a newline, monitor.h:28; tests/test_ref_pin.py holds a file the reference wrote).
// One file per env does not scale to millions of envs: MonitorCsvLog writes the files of a chosen subset of env indices,
// from the info maps HipVecEnv::infos() builds for the envs that just finished.
#pragma once
#include <any>
#include <filesystem>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "gym/vec_env.h"

namespace gym {

class EpisodeCsvWriter {   // the role of gym::ResultWriter (monitor.h:20-35)
public:
    EpisodeCsvWriter(std::filesystem::path file_name, size_t t_start, std::string const& id) {
        if (file_name.empty()) throw std::invalid_argument("EpisodeCsvWriter: empty file name");   // monitor.h:25 asserts
        out.open(file_name.replace_extension(".csv"));
        if (!out) throw std::runtime_error("EpisodeCsvWriter: cannot open " + file_name.string());
        out << "t_start: " << t_start << "\nid: " << id << "\n" << "t,r,l";
    }
    EpisodeCsvWriter& operator<<(Result const& w) { out << w.t << "," << w.r << "," << w.l << "\n"; return *this; }
    void flush() { out.flush(); }

private:
    std::ofstream out;
};

class MonitorCsvLog {
public:
    // one `<directory>/<prefix><global index>.csv` per entry of `env_indices` (local indices); env_offset = the shard's first global index
    MonitorCsvLog(std::filesystem::path const& directory, std::vector<int> const& env_indices, size_t t_start, int64_t env_offset = 0,
                  std::string const& prefix = "env") {
        std::filesystem::create_directories(directory);
        for (int i : env_indices) {
            const std::string id = prefix + std::to_string(env_offset + i);
            writers.emplace(i, EpisodeCsvWriter(directory / id, t_start, id));
        }
    }
    // call after every step with HipVecEnv::infos(): rows for the logged envs whose episode just finished (monitor.h:93-94)
    void log(std::vector<AnyMap> const& infos) {
        for (auto& [i, w] : writers) {
            if (i < 0 || (size_t)i >= infos.size()) continue;
            auto it = infos[i].find("episode");
            if (it != infos[i].end()) w << std::any_cast<Result>(it->second);
        }
    }
    void flush() { for (auto& [i, w] : writers) w.flush(); }

private:
    std::map<int, EpisodeCsvWriter> writers;
};

}  // namespace gym
