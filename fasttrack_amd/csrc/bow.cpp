// Frame::ComputeBoW (reference src/Frame.cc:762-769): DBoW2's TemplatedVocabulary::transform(features, BowVector&,
// FeatureVector&, levelsup) (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1194).  The tree walk of every
// descriptor - the cost of the call: k * L Hamming distances per feature - runs on the device (kernels_bow.hip); the
// two std::map results are assembled here in the reference's insertion order, so that the double sums of addWeight
// (BowVector.cpp:32-44) and of normalize (:60-85) associate exactly as on the CPU.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>

#include "ft_host.h"

#define FT_REQUIRE(cond, msg)               \
    do {                                    \
        if (!(cond)) {                      \
            ft_set_error(msg);              \
            return FT_ERR_INVALID;          \
        }                                   \
    } while (0)

struct ft_vocabulary {
    ft_context *ctx = nullptr;
    int k = 0, L = 0, scoring = 0, weighting = 0, nNodes = 0, nWords = 0;
    int *d_childStart = nullptr;
    unsigned *d_childList = nullptr, *d_wordId = nullptr;
    uint8_t *d_desc = nullptr;
    double *d_weight = nullptr;
    // per-call scratch (grow only; calls on one vocabulary are serialised by the mutex)
    std::mutex mutex;
    uint8_t *d_in = nullptr;
    unsigned *d_word = nullptr, *d_node = nullptr;
    double *d_w = nullptr;
    unsigned *h_word = nullptr, *h_node = nullptr;
    double *h_w = nullptr;
    int cap = 0;
};

namespace {

void freeScratch(ft_vocabulary *v) {
    hipFree(v->d_in);
    hipFree(v->d_word);
    hipFree(v->d_node);
    hipFree(v->d_w);
    hipHostFree(v->h_word);
    hipHostFree(v->h_node);
    hipHostFree(v->h_w);
    v->d_in = nullptr;
    v->d_word = v->d_node = nullptr;
    v->d_w = nullptr;
    v->h_word = v->h_node = nullptr;
    v->h_w = nullptr;
    v->cap = 0;
}

int ensureScratch(ft_vocabulary *v, int n) {
    if (n <= v->cap) return FT_OK;
    freeScratch(v);
    const int want = n + n / 2 + 64;
    FT_HIP(hipMalloc((void **)&v->d_in, (size_t)32 * want));
    FT_HIP(hipMalloc((void **)&v->d_word, sizeof(unsigned) * want));
    FT_HIP(hipMalloc((void **)&v->d_node, sizeof(unsigned) * want));
    FT_HIP(hipMalloc((void **)&v->d_w, sizeof(double) * want));
    FT_HIP(hipHostMalloc((void **)&v->h_word, sizeof(unsigned) * want, hipHostMallocDefault));
    FT_HIP(hipHostMalloc((void **)&v->h_node, sizeof(unsigned) * want, hipHostMallocDefault));
    FT_HIP(hipHostMalloc((void **)&v->h_w, sizeof(double) * want, hipHostMallocDefault));
    v->cap = want;
    return FT_OK;
}

}  // namespace

extern "C" {

int ft_vocabulary_create(ft_context *ctx, int k, int L, int scoring, int weighting, int n_nodes, const int *parent,
                         const uint8_t *is_leaf, const uint8_t *descriptors, const double *weights, ft_vocabulary **out) {
    FT_REQUIRE(ctx && out && parent && is_leaf && descriptors && weights, "ft_vocabulary_create: null argument");
    // the limits of the reference's loader (TemplatedVocabulary.h:1359)
    FT_REQUIRE(k >= 0 && k <= 20 && L >= 1 && L <= 10 && scoring >= 0 && scoring <= 5 && weighting >= 0 && weighting <= 3,
               "ft_vocabulary_create: k, L, scoring or weighting out of range");
    FT_REQUIRE(n_nodes >= 1, "ft_vocabulary_create: a vocabulary has at least its root");
    for (int i = 1; i < n_nodes; i++)
        FT_REQUIRE(parent[i] >= 0 && parent[i] < n_nodes && parent[i] != i, "ft_vocabulary_create: parent out of range");
    int rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    // children in ascending node id: the order loadFromTextFile's push_back produces (:1389-1390)
    std::vector<int> start(n_nodes + 1, 0);
    for (int i = 1; i < n_nodes; i++) start[parent[i] + 1]++;
    for (int i = 0; i < n_nodes; i++) start[i + 1] += start[i];
    std::vector<unsigned> list(std::max(n_nodes - 1, 1), 0u);
    {
        std::vector<int> cur(start.begin(), start.end() - 1);
        for (int i = 1; i < n_nodes; i++) list[cur[parent[i]]++] = (unsigned)i;
    }
    // words are the nodes flagged as leaves, numbered in node order (:1408-1416); a leaf has no children
    std::vector<unsigned> word(n_nodes, 0u);
    int nWords = 0;
    for (int i = 1; i < n_nodes; i++)
        if (is_leaf[i]) {
            FT_REQUIRE(start[i + 1] == start[i], "ft_vocabulary_create: a node flagged as leaf has children");
            word[i] = (unsigned)nWords++;
        } else {
            FT_REQUIRE(start[i + 1] > start[i], "ft_vocabulary_create: an inner node has no children");
        }
    ft_vocabulary *v = new ft_vocabulary;
    v->ctx = ctx;
    v->k = k;
    v->L = L;
    v->scoring = scoring;
    v->weighting = weighting;
    v->nNodes = n_nodes;
    v->nWords = nWords;
    hipError_t e = hipMalloc((void **)&v->d_childStart, sizeof(int) * (n_nodes + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_childList, sizeof(unsigned) * list.size());
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_wordId, sizeof(unsigned) * n_nodes);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_desc, (size_t)32 * n_nodes);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_weight, sizeof(double) * n_nodes);
    if (e == hipSuccess) e = hipMemcpy(v->d_childStart, start.data(), sizeof(int) * (n_nodes + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_childList, list.data(), sizeof(unsigned) * list.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_wordId, word.data(), sizeof(unsigned) * n_nodes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_desc, descriptors, (size_t)32 * n_nodes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_weight, weights, sizeof(double) * n_nodes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        ft_vocabulary_destroy(v);
        return ft_hip_fail(e, "vocabulary upload", __FILE__, __LINE__);
    }
    *out = v;
    return FT_OK;
}

int ft_vocabulary_load_text(ft_context *ctx, const char *path, ft_vocabulary **out) {
    FT_REQUIRE(ctx && path && out, "ft_vocabulary_load_text: null argument");
    std::ifstream f(path);
    FT_REQUIRE(f.good(), "ft_vocabulary_load_text: cannot open the file");
    std::string line;
    std::getline(f, line);
    int k = -1, L = -1, n1 = -1, n2 = -1;
    {
        std::stringstream ss(line);
        ss >> k >> L >> n1 >> n2;
    }
    FT_REQUIRE(!(k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3),
               "ft_vocabulary_load_text: not a vocabulary text file");  // :1359-1363
    std::vector<int> parent(1, 0);
    std::vector<uint8_t> leaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    while (std::getline(f, line)) {
        if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;  // the reference reads a trailing empty line as a node of the root; it is never reached by a walk
        std::stringstream ss(line);
        int pid = 0, isLeaf = 0;
        ss >> pid >> isLeaf;
        const size_t at = desc.size();
        desc.resize(at + 32, 0);
        for (int i = 0; i < 32; i++) {
            int b = 0;
            ss >> b;
            if (!ss.fail()) desc[at + i] = (uint8_t)b;  // FORB::fromString (FORB.cpp:120-135)
        }
        double w = 0.0;
        ss >> w;
        parent.push_back(pid);
        leaf.push_back(isLeaf > 0 ? 1 : 0);
        weight.push_back(w);
    }
    return ft_vocabulary_create(ctx, k, L, n1, n2, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), out);
}

int ft_vocabulary_destroy(ft_vocabulary *v) {
    if (!v) return FT_OK;
    ft_set_device(v->ctx);
    freeScratch(v);
    hipFree(v->d_childStart);
    hipFree(v->d_childList);
    hipFree(v->d_wordId);
    hipFree(v->d_desc);
    hipFree(v->d_weight);
    delete v;
    return FT_OK;
}

int ft_vocabulary_info(const ft_vocabulary *v, int *k, int *L, int *n_nodes, int *n_words) {
    FT_REQUIRE(v, "ft_vocabulary_info: null vocabulary");
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (n_nodes) *n_nodes = v->nNodes;
    if (n_words) *n_words = v->nWords;
    return FT_OK;
}

int ft_bow_transform(ft_vocabulary *v, const uint8_t *descriptors, int n, int on_device, int levelsup, unsigned *word_ids,
                     unsigned *node_ids, double *weights, unsigned *bow_ids, double *bow_values, int bow_capacity,
                     int *n_bow, unsigned *fv_nodes, int *fv_offsets, unsigned *fv_features, int fv_capacity, int *n_fv) {
    FT_REQUIRE(v && n >= 0 && (n == 0 || descriptors), "ft_bow_transform: bad argument");
    FT_REQUIRE(!bow_ids == !bow_values && (!bow_ids || n_bow), "ft_bow_transform: bow_ids, bow_values and n_bow go together");
    FT_REQUIRE(!fv_nodes == !fv_offsets && !fv_nodes == !fv_features && (!fv_nodes || n_fv),
               "ft_bow_transform: fv_nodes, fv_offsets, fv_features and n_fv go together");
    if (n_bow) *n_bow = 0;
    if (n_fv) *n_fv = 0;
    if (fv_offsets && fv_capacity >= 0) fv_offsets[0] = 0;
    if (n == 0 || v->nNodes <= 1) return FT_OK;  // empty() vocabulary: both results stay empty (:1134-1137)
    int rc = ft_set_device(v->ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(v->mutex);
    rc = ensureScratch(v, n);
    if (rc != FT_OK) return rc;
    hipStream_t st = v->ctx->stream;
    const uint8_t *d_desc = descriptors;
    if (!on_device) {
        FT_HIP(hipMemcpyAsync(v->d_in, descriptors, (size_t)32 * n, hipMemcpyHostToDevice, st));
        d_desc = v->d_in;
    }
    FtBowTree t;
    t.childStart = v->d_childStart;
    t.childList = v->d_childList;
    t.desc = v->d_desc;
    t.wordId = v->d_wordId;
    t.weight = v->d_weight;
    t.nNodes = v->nNodes;
    rc = ft_launch_bow_walk(st, t, d_desc, n, v->L - levelsup, v->d_word, v->d_node, v->d_w);
    if (rc != FT_OK) return rc;
    FT_HIP(hipMemcpyAsync(v->h_word, v->d_word, sizeof(unsigned) * n, hipMemcpyDeviceToHost, st));
    FT_HIP(hipMemcpyAsync(v->h_node, v->d_node, sizeof(unsigned) * n, hipMemcpyDeviceToHost, st));
    FT_HIP(hipMemcpyAsync(v->h_w, v->d_w, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    FT_HIP(hipStreamSynchronize(st));
    if (word_ids) memcpy(word_ids, v->h_word, sizeof(unsigned) * n);
    if (node_ids) memcpy(node_ids, v->h_node, sizeof(unsigned) * n);
    if (weights) memcpy(weights, v->h_w, sizeof(double) * n);

    // ---- BowVector (std::map<WordId, WordValue>): features in order, entries in ascending word id ----
    if (bow_ids) {
        // (word, feature) pairs of the features that are not stopped (w > 0, :1157 / :1185), stable by word
        std::vector<std::pair<unsigned, int>> order;
        order.reserve(n);
        for (int i = 0; i < n; i++)
            if (v->h_w[i] > 0) order.emplace_back(v->h_word[i], i);
        std::stable_sort(order.begin(), order.end(), [](const std::pair<unsigned, int> &a, const std::pair<unsigned, int> &b) { return a.first < b.first; });
        const bool tf = v->weighting == 0 || v->weighting == 1;  // TF_IDF or TF: addWeight; IDF / BINARY: addIfNotExist
        int m = 0;
        for (size_t p = 0; p < order.size();) {
            size_t q = p;
            double val = v->h_w[order[p].second];  // the insert of the first feature of this word
            for (q = p + 1; q < order.size() && order[q].first == order[p].first; q++)
                if (tf) val += v->h_w[order[q].second];  // vit->second += v, in feature order
            if (m >= bow_capacity) {
                ft_set_error("ft_bow_transform: bow_capacity too small");
                return FT_ERR_CAPACITY;
            }
            bow_ids[m] = order[p].first;
            bow_values[m] = val;
            m++;
            p = q;
        }
        const bool must = v->scoring != 5;  // every scoring but DOT_PRODUCT normalises (ScoringObject.h:73-89)
        if (tf && m > 0 && !must) {          // :1164-1170
            const double nd = (double)m;
            for (int j = 0; j < m; j++) bow_values[j] /= nd;
        }
        if (must) {  // BowVector::normalize (BowVector.cpp:60-85): L2 for L2_NORM, L1 otherwise
            double norm = 0.0;
            if (v->scoring != 1) {
                for (int j = 0; j < m; j++) norm += fabs(bow_values[j]);
            } else {
                for (int j = 0; j < m; j++) norm += bow_values[j] * bow_values[j];
                norm = sqrt(norm);
            }
            if (norm > 0.0)
                for (int j = 0; j < m; j++) bow_values[j] /= norm;
        }
        *n_bow = m;
    }
    // ---- FeatureVector (std::map<NodeId, std::vector<unsigned>>) in CSR form ----
    if (fv_nodes) {
        std::vector<std::pair<unsigned, int>> order;
        order.reserve(n);
        for (int i = 0; i < n; i++)
            if (v->h_w[i] > 0) order.emplace_back(v->h_node[i], i);
        std::stable_sort(order.begin(), order.end(), [](const std::pair<unsigned, int> &a, const std::pair<unsigned, int> &b) { return a.first < b.first; });
        int m = 0;
        for (size_t p = 0; p < order.size(); p++) {
            if (p == 0 || order[p].first != order[p - 1].first) {
                if (m >= fv_capacity) {
                    ft_set_error("ft_bow_transform: fv_capacity too small");
                    return FT_ERR_CAPACITY;
                }
                fv_nodes[m] = order[p].first;
                fv_offsets[m] = (int)p;
                m++;
            }
            fv_features[p] = (unsigned)order[p].second;
        }
        fv_offsets[m] = (int)order.size();
        *n_fv = m;
    }
    return FT_OK;
}

}  // extern "C"

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:322-524): the matching inside the common
// nodes on the device (k_search_by_bow), the rotation-consistency filter (:489-521, ComputeThreeMaxima :2210-2251) here on the
// downloaded assignments - the order inside a histogram bin does not matter to it, only the bins' sizes.
namespace {
int checkBowSide(const ft_bow_side *s, const char *who) {
    auto bad = [&](const char *what) {
        ft_set_error(std::string("ft_search_by_bow: ") + who + ": " + what);
        return FT_ERR_INVALID;
    };
    if (s->n < 0 || s->n >= (1 << 20) || s->n_nodes < 0) return bad("count out of range");
    if (s->n_nodes > 0 && (!s->fv_nodes || !s->fv_offsets)) return bad("null feature vector");
    if (s->n > 0 && !s->descriptors) return bad("null descriptors");
    if (s->n_nodes == 0) return FT_OK;
    if (s->fv_offsets[0] != 0) return bad("fv_offsets[0] != 0");
    for (int j = 0; j < s->n_nodes; j++) {
        if (s->fv_offsets[j + 1] < s->fv_offsets[j]) return bad("fv_offsets not ascending");
        if (j > 0 && s->fv_nodes[j] <= s->fv_nodes[j - 1]) return bad("fv_nodes not strictly ascending");
    }
    const int total = s->fv_offsets[s->n_nodes];
    if (total > s->n) return bad("more feature-vector entries than features");
    if (total > 0 && !s->fv_features) return bad("null fv_features");
    for (int e = 0; e < total; e++)
        if (s->fv_features[e] >= (unsigned)s->n) return bad("feature index out of range");
    return FT_OK;
}
}  // namespace

int ft_search_by_bow(ft_context *ctx, const ft_bow_side *kf, const uint8_t *kf_has_point, const ft_bow_side *frame, int frame_nleft,
                     float nn_ratio, int check_orientation, int *matches, int *n_matches) {
    FT_REQUIRE(ctx && kf && frame && matches, "ft_search_by_bow: null argument");
    FT_REQUIRE(kf->n == 0 || kf_has_point, "ft_search_by_bow: null kf_has_point");
    FT_REQUIRE(frame_nleft >= -1 && frame_nleft <= frame->n, "ft_search_by_bow: frame_nleft out of range");
    int rc = checkBowSide(kf, "keyframe");
    if (rc == FT_OK) rc = checkBowSide(frame, "frame");
    if (rc != FT_OK) return rc;
    FT_REQUIRE(!check_orientation || ((kf->n == 0 || kf->angles) && (frame->n == 0 || frame->angles)),
               "ft_search_by_bow: check_orientation needs the keypoint angles of both sides");
    if (n_matches) *n_matches = 0;
    const int N = frame->n;
    for (int i = 0; i < N; i++) matches[i] = -1;
    if (N == 0 || kf->n == 0 || kf->n_nodes == 0 || frame->n_nodes == 0) return FT_OK;
    rc = ft_set_device(ctx);
    if (rc != FT_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->matchMutex);
    // one block of the context's scratch: per side nodes | offsets | features | descriptors, then has_point, then matches
    auto up64 = [](size_t v) { return (v + 63) & ~(size_t)63; };
    struct Lay { size_t nodes, offsets, features, desc; };
    size_t off = 0;
    auto lay = [&](const ft_bow_side *s) {
        Lay L;
        L.nodes = off; off += up64(sizeof(unsigned) * (size_t)s->n_nodes);
        L.offsets = off; off += up64(sizeof(int) * ((size_t)s->n_nodes + 1));
        L.features = off; off += up64(sizeof(unsigned) * (size_t)std::max(s->fv_offsets[s->n_nodes], 1));
        L.desc = off; off += up64((size_t)32 * s->n);
        return L;
    };
    const Lay lk_ = lay(kf), lf = lay(frame);
    const size_t oHas = off; off += up64((size_t)kf->n);
    const size_t oMatches = off; off += up64(sizeof(int) * (size_t)N);
    rc = ft_ensure_scratch(ctx, off, off);
    if (rc != FT_OK) return rc;
    uint8_t *dev = (uint8_t *)ctx->scratchDev, *pin = (uint8_t *)ctx->scratchPin;
    hipStream_t st = ctx->stream;
    auto fill = [&](const ft_bow_side *s, const Lay &L) {
        memcpy(pin + L.nodes, s->fv_nodes, sizeof(unsigned) * (size_t)s->n_nodes);
        memcpy(pin + L.offsets, s->fv_offsets, sizeof(int) * ((size_t)s->n_nodes + 1));
        memcpy(pin + L.features, s->fv_features, sizeof(unsigned) * (size_t)s->fv_offsets[s->n_nodes]);
        memcpy(pin + L.desc, s->descriptors, (size_t)32 * s->n);
    };
    fill(kf, lk_);
    fill(frame, lf);
    memcpy(pin + oHas, kf_has_point, (size_t)kf->n);
    memset(pin + oMatches, 0xff, sizeof(int) * (size_t)N);
    FT_HIP(hipMemcpyAsync(dev, pin, off, hipMemcpyHostToDevice, st));  // one copy: the block is contiguous
    auto side = [&](const ft_bow_side *s, const Lay &L) {
        FtBowSide D;
        D.n = s->n;
        D.nNodes = s->n_nodes;
        D.nodes = (const unsigned *)(dev + L.nodes);
        D.offsets = (const int *)(dev + L.offsets);
        D.features = (const unsigned *)(dev + L.features);
        D.desc = dev + L.desc;
        return D;
    };
    rc = ft_launch_search_by_bow(st, side(kf, lk_), dev + oHas, side(frame, lf), frame_nleft, nn_ratio, (int *)(dev + oMatches));
    if (rc != FT_OK) return rc;
    FT_HIP(hipMemcpyAsync(pin + oMatches, dev + oMatches, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost, st));
    FT_HIP(hipStreamSynchronize(st));
    memcpy(matches, pin + oMatches, sizeof(int) * (size_t)N);
    int nm = 0;
    for (int i = 0; i < N; i++) nm += matches[i] >= 0;
    if (check_orientation) {
        std::vector<int> rotHist[FT_HISTO_LENGTH];
        const float factor = 1.0f / FT_HISTO_LENGTH;
        for (int i = 0; i < N; i++) {
            if (matches[i] < 0) continue;
            float rot = kf->angles[matches[i]] - frame->angles[i];
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == FT_HISTO_LENGTH) bin = 0;
            if (bin >= 0 && bin < FT_HISTO_LENGTH) rotHist[bin].push_back(i);  // the reference asserts
        }
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < FT_HISTO_LENGTH; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) {
                max3 = max2; max2 = max1; max1 = sz;
                ind3 = ind2; ind2 = ind1; ind1 = i;
            } else if (sz > max2) {
                max3 = max2; max2 = sz;
                ind3 = ind2; ind2 = i;
            } else if (sz > max3) {
                max3 = sz; ind3 = i;
            }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < FT_HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) {
                    matches[idx] = -1;
                    nm--;
                }
    }
    if (n_matches) *n_matches = nm;
    return FT_OK;
}

