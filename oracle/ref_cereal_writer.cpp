// oracle/ref_cereal_writer.cpp -- TEST INFRASTRUCTURE: writes a detection_model-shaped file with the
// REFERENCE'S OWN cereal-1.1.1 (compiled from /root/reference/3rdparty/cereal-1.1.1/include by oracle/Makefile, nothing
// copied) so that the product's model file writer / reader (include/sdm_io/binary_archive.hpp, model_io.py) is pinned
// byte for byte to what cereal::BinaryOutputArchive really emits.
//
// The reference's serialisable types need OpenCV (cv::Mat), which is not in the tree; the stand-ins below repeat the
// reference's serialize()/save() member lists VERBATIM IN ORDER (cited) on plain structs, so every byte of the output --
// vector sizes, string records, enum encodings, bools, the Mat record with cereal::binary_data -- is produced by real
// cereal code.  Values: a fixed, seeded model (2 levels, 3 landmarks); tests/golden/make_golden_cereal.py stores the
// output as tests/golden/cereal_ref_model.bin, tests/test_model_file_pinning.py compares.
#include <cereal/archives/binary.hpp>
#include <cereal/types/string.hpp>
#include <cereal/types/vector.hpp>

#include <fstream>
#include <string>
#include <vector>

struct MatRec {      // cv::Mat, CV_32FC1 = 5
    int rows = 0, cols = 0, type = 5;
    bool continuous = true;
    std::vector<float> data;
};
// include/superviseddescent/utils/mat_cerealisation.hpp:42-58 (the continuous branch; cv::Mat::create'd matrices are continuous)
template <class Archive>
void save(Archive& ar, const MatRec& mat)
{
    int rows = mat.rows, cols = mat.cols, type = mat.type;
    bool continuous = mat.continuous;
    ar & rows & cols & type & continuous;
    const int data_size = rows * cols * 4;
    auto mat_data = cereal::binary_data(mat.data.data(), data_size);
    ar & mat_data;
}

struct Regulariser {      // include/superviseddescent/regressors.hpp:87-169
    enum class RegularisationType { Manual, MatrixNorm };      // :96-98
    RegularisationType regularisation_type = RegularisationType::Manual;
    float lambda = 0.0f;
    bool regularise_last_row = true;
    template <class Archive> void serialize(Archive& ar) { ar(regularisation_type, lambda, regularise_last_row); }      // :165-168
};
struct LinearRegressor {      // regressors.hpp:318-400
    MatRec x;
    Regulariser regulariser;
    template <class Archive> void serialize(Archive& ar) { ar(x, regulariser); }      // :396-399
};
struct InterEyeDistanceNormalisation {      // include/rcr/model.hpp:84-116
    std::vector<std::string> modelLandmarksList, rightEyeIdentifiers, leftEyeIdentifiers;
    template <class Archive> void serialize(Archive& archive) { archive(modelLandmarksList, rightEyeIdentifiers, leftEyeIdentifiers); }      // :111-115
};
struct SupervisedDescentOptimiser {      // include/superviseddescent/superviseddescent.hpp:85-361
    std::vector<LinearRegressor> regressors;
    InterEyeDistanceNormalisation normalisation_strategy;
    template <class Archive> void serialize(Archive& ar) { ar(regressors, normalisation_strategy); }      // :356-360
};
enum VlHogVariant_ { VlHogVariantDalalTriggs, VlHogVariantUoctti };      // include/rcr/hog.h:72
struct HoGParam {      // include/rcr/adaptive_vlhog.hpp:41-60
    VlHogVariant_ vlhog_variant;
    int num_cells, cell_size, num_bins;
    float relative_patch_size;
    template <class Archive> void serialize(Archive& ar) { ar(vlhog_variant, num_cells, cell_size, num_bins, relative_patch_size); }      // :55-59
};
struct detection_model {      // include/rcr/model.hpp:122-183
    SupervisedDescentOptimiser optimised_model;
    MatRec mean;
    std::vector<std::string> landmark_ids;
    std::vector<HoGParam> hog_params;
    std::vector<std::string> right_eye_ids, left_eye_ids;
    template <class Archive> void serialize(Archive& archive) { archive(optimised_model, mean, landmark_ids, hog_params, right_eye_ids, left_eye_ids); }      // :178-182
};

int main(int argc, char** argv)
{
    if (argc != 2) return 2;
    detection_model m;
    const std::vector<std::string> ids{"37", "40", "9"}, re{"37"}, le{"40"};
    for (int l = 0; l < 2; ++l) {
        LinearRegressor r;
        r.x.rows = 3; r.x.cols = 4;
        for (int i = 0; i < 12; ++i) r.x.data.push_back(0.25f * i + l);
        r.regulariser.regularisation_type = l == 0 ? Regulariser::RegularisationType::MatrixNorm : Regulariser::RegularisationType::Manual;
        r.regulariser.lambda = l == 0 ? 1.5f : 0.125f;
        r.regulariser.regularise_last_row = l == 1;
        m.optimised_model.regressors.push_back(r);
    }
    m.optimised_model.normalisation_strategy = {ids, re, le};
    m.mean.rows = 1; m.mean.cols = 6;
    m.mean.data = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f};
    m.landmark_ids = ids;
    m.hog_params = {{VlHogVariantUoctti, 5, 11, 4, 1.0f}, {VlHogVariantDalalTriggs, 3, 10, 9, 0.7f}};
    m.right_eye_ids = re;
    m.left_eye_ids = le;
    std::ofstream file(argv[1], std::ios::binary);
    cereal::BinaryOutputArchive output_archive(file);      // model.hpp:214-218 (save_detection_model)
    output_archive(m);
    return file ? 0 : 1;
}
