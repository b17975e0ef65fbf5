// spaces_probe.h — one probe, two subjects.  Prints, as JSON, what a tree's space types report: Discrete / Box construction
// (sizes, types, ranges, exceptions), make_shape<T>'s dtype mapping, clone / as<T> / keys, and Env::seed's effect on the env's
// generator.  Compiled (a) by oracle/Makefile `_ref` against the REFERENCE's own spaces/*.h, spaces/*.cpp and env.h where they
// lie — that output is frozen as tests/golden/ref_spaces.json — and (b) by tests/cpp/Makefile against the stand-alone mirror
// gym/vec_env.h, whose output must equal the frozen file (tests/test_ref_pin.py).  Uses only names both trees define.
#pragma once
#include <cstdio>
#include <random>
#include <sstream>
#include <string>
#include <vector>

namespace spaces_probe {

inline std::string vec(std::vector<int64_t> const& v) {
    std::ostringstream o; o << "[";
    for (size_t i = 0; i < v.size(); i++) o << (i ? "," : "") << v[i];
    o << "]"; return o.str();
}
template <class T> std::string ranges(std::vector<gym::space::Range<T>> const& r) {
    std::ostringstream o; o.precision(9); o << "[";
    for (size_t i = 0; i < r.size(); i++) o << (i ? "," : "") << "[" << (double)r[i].low << "," << (double)r[i].high << "]";
    o << "]"; return o.str();
}
template <class F> std::string thrown(F&& f) {
    try { f(); } catch (std::invalid_argument const&) { return "\"invalid_argument\""; } catch (std::runtime_error const&) { return "\"runtime_error\""; }
    catch (std::exception const&) { return "\"exception\""; }
    return "null";
}

// An env that exposes the generator Env::seed() seeds (env.h:60-68, :83) the way CartPoleEnv::reset draws from it (cartpole.cpp:21-26)
struct SeedProbeEnv : gym::Env<std::vector<double>, int> {
    std::vector<double> reset() noexcept override { return {}; }
    StepT step(const int&) noexcept override { return {}; }
    uint32_t raw() { return (uint32_t)m_Device(); }
    float draw() { std::uniform_real_distribution<float> d(-0.05f, 0.05f); return d(m_Device); }
};

inline std::string run() {
    using namespace gym::space;
    std::ostringstream o; o.precision(9);
    o << "{\n";
    {
        Discrete d(5);
        o << " \"discrete5\": {\"size\": " << vec(d.size()) << ", \"type\": " << (int)d.type() << ", \"n\": " << d.n << ", \"keys\": \"" << d.keys()[0] << "\"},\n";
        bool in_range = true;
        for (int k = 0; k < 2000; k++) { int v = d.sample<int>(); in_range = in_range && v >= 0 && v < 5; }
        o << " \"discrete5_samples_in_range\": " << (in_range ? "true" : "false") << ",\n";
        auto c = d.clone();
        o << " \"discrete5_clone\": {\"is_discrete\": " << (c->as<Discrete>() ? "true" : "false") << ", \"is_box_float\": " << (c->as<Box<float>>() ? "true" : "false")
          << ", \"n\": " << c->as<Discrete>()->n << "},\n";
    }
    o << " \"discrete0_throws\": " << thrown([] { Discrete d(0); }) << ",\n";
    o << " \"discrete_neg_throws\": " << thrown([] { Discrete d(-3); }) << ",\n";
    {
        Box<float> b(Shape{{2}, c10::kFloat});                                        // mountain_car.cpp:11 shape, default range (box.h:31-34)
        o << " \"box_default_float\": {\"size\": " << vec(b.size()) << ", \"type\": " << (int)b.type() << ", \"ranges\": " << ranges(b.getRange()) << "},\n";
        Box<int> bi(Shape{{3}, c10::kInt});
        o << " \"box_default_int\": {\"ranges\": " << ranges(bi.getRange()) << "},\n";
    }
    {
        std::vector<Range<float>> r{{-4.8f, 4.8f}, {-3.4028235e38f, 3.4028235e38f}, {-0.41887903f, 0.41887903f}, {-1.f, 2.f}};
        Box<float> b(r, make_shape<float>(4));                                         // cartpole.cpp:10-16 form
        o << " \"box_vector_float\": {\"size\": " << vec(b.size()) << ", \"type\": " << (int)b.type() << ", \"ranges\": " << ranges(b.getRange()) << "},\n";
        std::vector<float> lo, hi; b.getRange(lo, hi);
        o << " \"box_vector_float_lows_highs\": [" << lo.size() << "," << hi.size() << "," << (double)lo[3] << "," << (double)hi[3] << "],\n";
        auto c = b.clone();
        o << " \"box_clone\": {\"is_box_float\": " << (c->as<Box<float>>() ? "true" : "false") << ", \"is_discrete\": " << (c->as<Discrete>() ? "true" : "false")
          << ", \"ranges\": " << ranges(c->as<Box<float>>()->getRange()) << "},\n";
        c->setName("pixels");
        o << " \"set_name\": \"" << c->keys()[0] << "\",\n";
        auto smp = b.sample<std::vector<float>>();
        o << " \"box_sample_len\": " << smp.size() << ",\n";
    }
    o << " \"box_vector_zero_dim_throws\": " << thrown([] { Box<float> b(std::vector<Range<float>>{{0.f, 1.f}}, Shape{{}, c10::kFloat}); }) << ",\n";
    o << " \"box_vector_two_dim_throws\": " << thrown([] { Box<float> b(std::vector<Range<float>>{{0.f, 1.f}}, Shape{{2, 2}, c10::kFloat}); }) << ",\n";
    o << " \"box_vector_empty_range_throws\": " << thrown([] { Box<float> b(std::vector<Range<float>>{}, Shape{{4}, c10::kFloat}); }) << ",\n";
    {
        Box<uint8_t> img(Range<uint8_t>{0, 255}, make_shape<uint8_t>(std::vector<int64_t>{84, 84, 1}));   // warp_frame.h:28
        o << " \"box_image_u8\": {\"size\": " << vec(img.size()) << ", \"type\": " << (int)img.type() << ", \"ranges\": " << ranges(img.getRange()) << "},\n";
    }
    o << " \"make_shape_types\": {\"int\": " << (int)make_shape<int>(3).m_Type << ", \"uint8\": " << (int)make_shape<uint8_t>(3).m_Type << ", \"float\": "
      << (int)make_shape<float>(3).m_Type << ", \"double\": " << (int)make_shape<double>(3).m_Type << ", \"bool\": " << (int)make_shape<bool>(3).m_Type
      << ", \"uint8_nd\": " << (int)make_shape<uint8_t>(std::vector<int64_t>{2, 3}).m_Type << ", \"float_nd_axis\": " << vec(make_shape<float>(std::vector<int64_t>{2, 3}).m_Axis) << "},\n";
    o << " \"dim\": [" << dim(Shape{{}, c10::kInt}) << "," << dim(Shape{{7}, c10::kInt}) << "," << dim(Shape{{7, 8, 9}, c10::kInt}) << "],\n";
    {
        // Env::seed (env.h:60-68): m_Device.seed(value); an env that was never seeded keeps the default-constructed generator (env.h:83)
        o << " \"env_seed\": {";
        SeedProbeEnv unseeded;
        o << "\"unseeded_raw\": [" << unseeded.raw() << "," << unseeded.raw() << "]";
        for (uint64_t s : {0ull, 1ull, 7ull, 12345ull, 4294967296ull + 5ull}) {
            SeedProbeEnv e; e.seed(s);
            o << ", \"raw_" << s << "\": [" << e.raw() << "," << e.raw() << "," << e.raw() << "," << e.raw() << "]";
            SeedProbeEnv e2; e2.seed(s);
            o << ", \"draws_" << s << "\": [";
            for (int k = 0; k < 8; k++) { o << (k ? "," : "") << (double)e2.draw(); }
            o << "]";
        }
        o << "}\n";
    }
    o << "}\n";
    return o.str();
}

}  // namespace spaces_probe
