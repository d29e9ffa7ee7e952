// FileHandler.h -- the on-disk formats either side of the optimiser, byte-compatible
// with the reference's writers so that its plotting / comparison scripts keep working:
//   savedTrajecInfo<prefix>/{A_matrices,B_matrices,states,controls}.csv  FileHandler::SaveTrajecInformation
//                                                                       (src/FileHandler/FileHandler.cpp:312-383)
//   savedTrajecInfo<prefix>/keypoints.csv                               FileHandler::SaveKeypointsToFile (:385-420)
//   TestTasks/<task>/<i>.csv                                            SaveTaskToFile / LoadTaskFromFile (:422-578)
//   <method>/summary.csv                                                GenTestingData (src/GenTestingData.cpp:229-256)
// Numbers go through a default-formatted std::ostream (6 significant digits) and every value is followed by a
// comma, exactly as the reference writes them.
#pragma once
#include <string>
#include <vector>
#include "Matrix.h"

class FileHandler {
public:
    // One line per step for steps 0 .. horizon-2 (the reference stops at horizon-1, :331,347,361,373); matrices
    // row-major within a line.  Creates `root_dir`.  false when the directory cannot be created.
    static bool SaveTrajecInformation(const std::vector<MatrixXd> &A_matrices, const std::vector<MatrixXd> &B_matrices,
                                      const std::vector<MatrixXd> &states, const std::vector<MatrixXd> &controls,
                                      const std::string &root_dir);
    // One line per DoF listing the steps at which it is a key-point; dof = keypoints[0].size() as in the reference.
    static bool SaveKeypointsToFile(const std::string &root_dir, const std::vector<std::vector<int>> &keypoints);
    // Task row: start values (robot joints, body poses) then the residual targets, one line.
    static bool SaveTaskToFile(const std::string &filename, const std::vector<double> &start, const std::vector<double> &targets);
    // false (with a message on stderr) when the file is missing or does not hold n_start + n_targets values --
    // the reference exits the program in both cases (:483-486, :527-531).
    static bool LoadTaskFromFile(const std::string &filename, int n_start, int n_targets, std::vector<double> &start,
                                 std::vector<double> &targets);
    struct SummaryRow {
        double cost_reduction, optimisation_time_ms; int num_iterations; double avg_num_dofs, avg_percent_derivs;
        std::vector<double> time_derivs_ms, time_bp_ms, time_fp_ms;     // per-iteration timings of that run
    };
    // The columns headed "Average time ..." hold std::accumulate(begin, end, 0): an INT-typed running sum of the
    // per-iteration times, not an average (src/GenTestingData.cpp:229-231); reproduced as is.
    static bool SaveSummary(const std::string &filename, const std::vector<SummaryRow> &rows);
    static int IntAccumulate(const std::vector<double> &v);
};
