// ORACLE (test infrastructure, NOT product code) -- restatement of the part of the reference VM that
// produces execution traces for the prover benchmark inputs (Span / Group programs).
//
// Follows: /root/reference/src/utils/sponge.rs:13-65 (accumulator round), src/utils/hasher.rs:12-90 (RESCR round),
// src/processor/opcodes.rs:5-86 (encodings), src/programs/assembly/mod.rs:19-48,133-199,253-290 (compile, parse_branch,
// add_span, repeat), src/programs/assembly/parsers.rs:44-62 (push alignment), src/programs/blocks/mod.rs:88-218
// (Span, Group hash), src/programs/hashing.rs:15-74, src/programs/mod.rs:35-55 (program hash),
// src/processor/mod.rs:23-143 (execute), src/processor/decoder/mod.rs (trace builder), src/processor/stack/mod.rs.
// Switch / Loop blocks and the ops outside the subset below are out of scope for the oracle (they are not needed to
// build inputs of the prover hot path) and abort.
#pragma once
#include "field.hpp"
#include "rescue_constants.hpp"
#include <string>
#include <sstream>
#include <map>
#include <memory>
#include <stdexcept>

namespace orc {

static inline u128 cst(const uint64_t v[2]) { return make_u128(v[1], v[0]); }

static const size_t BASE_CYCLE_LENGTH = 16;     // lib.rs:85
static const size_t SPONGE_WIDTH = 4;           // lib.rs:104
static const size_t HASH_STATE_WIDTH = 6;       // lib.rs:98
static const size_t HACC_NUM_ROUNDS = 14;       // lib.rs:106
static const size_t MIN_TRACE_LENGTH = 16;      // lib.rs:82
static const size_t MIN_STACK_DEPTH = 8, MIN_CONTEXT_DEPTH = 1, MIN_LOOP_DEPTH = 1;   // lib.rs:87-89
static const size_t MAX_CONTEXT_DEPTH = 16, MAX_LOOP_DEPTH = 8, MAX_STACK_DEPTH = 32; // lib.rs:80,81,138
static const size_t MAX_PUBLIC_INPUTS = 8, MAX_REGISTER_COUNT = 128;                  // lib.rs:136,83
static const size_t PROGRAM_DIGEST_SIZE = 2;    // lib.rs:105
static const u128 RESCUE_INV_ALPHA = make_u128(0xAAAAAAAAAAAAAAAAull, 0xAAAA8CAAAAAAAAABull);  // sponge.rs:70

// ---- Rescue building blocks (generic width) -----------------------------------------------------------
template <size_t W> struct Rescue {
    static void sbox(u128* s) { for (size_t i = 0; i < W; i++) s[i] = exp(s[i], 3); }
    static void inv_sbox(u128* s) { for (size_t i = 0; i < W; i++) s[i] = exp(s[i], RESCUE_INV_ALPHA); }
    static void matmul(u128* s, const uint64_t (*m)[2]) {
        u128 r[W];
        for (size_t i = 0; i < W; i++) {
            u128 acc = 0;
            for (size_t j = 0; j < W; j++) acc = add(acc, mul(cst(m[i * W + j]), s[j]));
            r[i] = acc;
        }
        for (size_t i = 0; i < W; i++) s[i] = r[i];
    }
};
static inline void sponge_mds(u128* s) { Rescue<4>::matmul(s, SPONGE_MDS); }           // sponge.rs:52
static inline void sponge_inv_mds(u128* s) { Rescue<4>::matmul(s, SPONGE_INV_MDS); }   // sponge.rs:66 (apply_inv_mds)
static inline void hasher_mds(u128* s) { Rescue<6>::matmul(s, HASHER_MDS); }
static inline void hasher_inv_mds(u128* s) { Rescue<6>::matmul(s, HASHER_INV_MDS); }

// modified Rescue round with op_code / op_value injection                 sponge.rs:13-30
static inline void sponge_apply_round(u128* state, u128 op_code, u128 op_value, size_t step) {
    size_t idx = step % BASE_CYCLE_LENGTH;
    for (size_t i = 0; i < 4; i++) state[i] = add(state[i], cst(SPONGE_ARK[i][idx]));
    Rescue<4>::sbox(state);
    sponge_mds(state);
    state[0] = add(state[0], op_code);
    state[1] = add(state[1], op_value);
    for (size_t i = 0; i < 4; i++) state[i] = add(state[i], cst(SPONGE_ARK[4 + i][idx]));
    Rescue<4>::inv_sbox(state);
    sponge_mds(state);
}
// plain Rescue round on the 6-element hasher state                         hasher.rs:28-40
static inline void hasher_apply_round(u128* state, size_t step) {
    size_t idx = step % BASE_CYCLE_LENGTH;
    for (size_t i = 0; i < 6; i++) state[i] = add(state[i], cst(HASHER_ARK[i][idx]));
    Rescue<6>::sbox(state);
    hasher_mds(state);
    for (size_t i = 0; i < 6; i++) state[i] = add(state[i], cst(HASHER_ARK[6 + i][idx]));
    Rescue<6>::inv_sbox(state);
    hasher_mds(state);
}

// ---- opcodes (processor/opcodes.rs:5-86) ----------------------------------------------------------------
enum FlowOp : uint8_t { F_HACC = 0, F_BEGIN = 1, F_TEND = 2, F_FEND = 3, F_LOOP = 4, F_WRAP = 5, F_BREAK = 6, F_VOID = 7 };
enum UserOp : uint8_t {
    OP_ASSERT = 0x60, OP_ASSERTEQ = 0x61, OP_EQ = 0x62, OP_DROP = 0x63, OP_DROP4 = 0x64, OP_CHOOSE = 0x65, OP_CHOOSE2 = 0x66, OP_CSWAP2 = 0x67,
    OP_ADD = 0x68, OP_MUL = 0x69, OP_AND = 0x6A, OP_OR = 0x6B, OP_INV = 0x6C, OP_NEG = 0x6D, OP_NOT = 0x6E,
    OP_READ = 0x70, OP_READ2 = 0x71, OP_DUP = 0x72, OP_DUP2 = 0x73, OP_DUP4 = 0x74, OP_PAD2 = 0x75,
    OP_SWAP = 0x78, OP_SWAP2 = 0x79, OP_SWAP4 = 0x7A, OP_ROLL4 = 0x7B, OP_ROLL8 = 0x7C, OP_BINACC = 0x7D,
    OP_PUSH = 0x1F, OP_CMP = 0x3F, OP_RESCR = 0x5F, OP_BEGIN = 0x00, OP_NOOP = 0x7F,
};
static inline size_t ld_index(UserOp op) { return (size_t)op & 0x1F; }          // opcodes.rs:90
static inline size_t hd_index(UserOp op) { return ((size_t)op >> 5) & 3; }      // opcodes.rs:101

// ---- program blocks ---------------------------------------------------------------------------------------
struct Block;
struct Span { std::vector<UserOp> ops; std::map<size_t, u128> push_values; };
struct Block {
    bool is_span = true;
    Span span;
    std::vector<Block> body;    // Group
};

static inline void hash_op(u128* state, uint8_t op_code, u128 op_value, size_t step) { sponge_apply_round(state, op_code, op_value, step); }  // hashing.rs:62
static inline std::array<u128, 4> hash_acc(u128 parent_hash, u128 v0, u128 v1) {      // hashing.rs:68
    std::array<u128, 4> st = {parent_hash, v0, v1, 0};
    for (size_t i = 1; i < 1 + HACC_NUM_ROUNDS; i++) hash_op(st.data(), OP_NOOP, 0, i);
    return st;
}
static inline void span_hash(const Span& s, u128* state) {                          // blocks/mod.rs:149
    for (size_t i = 0; i < s.ops.size(); i++) {
        u128 v = 0;
        if (s.ops[i] == OP_PUSH) v = s.push_values.at(i);
        hash_op(state, s.ops[i], v, i);
    }
}
static u128 hash_seq(const std::vector<Block>& blocks);                               // hashing.rs:15
static inline std::pair<u128, u128> group_hash(const Block& g) { return {hash_seq(g.body), 0}; }   // blocks/mod.rs:211-218
static u128 hash_seq(const std::vector<Block>& blocks) {
    u128 st[4] = {0, 0, 0, 0};
    if (!blocks[0].is_span) throw std::runtime_error("first block in a sequence must be a Span block");
    span_hash(blocks[0].span, st);
    for (size_t b = 1; b < blocks.size(); b++) {
        if (blocks[b].is_span) {
            hash_op(st, OP_NOOP, 0, BASE_CYCLE_LENGTH - 1);
            span_hash(blocks[b].span, st);
        } else {
            auto h = group_hash(blocks[b]);
            auto m = hash_acc(st[0], h.first, h.second);
            for (int i = 0; i < 4; i++) st[i] = m[i];
        }
    }
    hash_op(st, OP_NOOP, 0, BASE_CYCLE_LENGTH - 1);    // BLOCK_SUFFIX at BLOCK_SUFFIX_OFFSET (blocks/mod.rs:9-10)
    return st[0];
}

struct Program {
    Block root;                  // Group
    u128 hash[2];
    void finalize() {            // programs/mod.rs:35-55
        if (!root.body[0].is_span || root.body[0].span.ops[0] != OP_BEGIN) throw std::runtime_error("a program must start with BEGIN operation");
        auto h = group_hash(root);
        auto acc = hash_acc(0, h.first, h.second);
        hash[0] = acc[0]; hash[1] = acc[1];
    }
};

// ---- assembly subset -----------------------------------------------------------------------------------------
struct Assembler {
    std::vector<std::string> tokens;

    static void add_span(std::vector<Block>& body, std::vector<UserOp>& ops, std::map<size_t, u128>& hints, bool force) {  // assembly/mod.rs:253
        if (ops.empty() && !force) return;
        Block b; b.is_span = true; b.span.ops = ops;
        size_t pad = BASE_CYCLE_LENGTH - (ops.size() % BASE_CYCLE_LENGTH) - 1;
        b.span.ops.resize(ops.size() + pad, OP_NOOP);
        b.span.push_values = hints;
        body.push_back(b);
        ops.clear(); hints.clear();
    }
    static Block merge_spans(const Block& a, const Block& b) {             // blocks/mod.rs:163
        Block r; r.is_span = true;
        r.span.ops = a.span.ops;
        r.span.ops.push_back(OP_NOOP);
        r.span.ops.insert(r.span.ops.end(), b.span.ops.begin(), b.span.ops.end());
        r.span.push_values = a.span.push_values;
        size_t off = a.span.ops.size() + 1;
        for (auto& kv : b.span.push_values) r.span.push_values[kv.first + off] = kv.second;
        return r;
    }
    static std::vector<Block> repeat_block_sequence(const std::vector<Block>& tpl, size_t n) {   // assembly/mod.rs:271
        std::vector<Block> body;
        if (!tpl.back().is_span) { for (size_t i = 0; i < n; i++) body.insert(body.end(), tpl.begin(), tpl.end()); }
        else {
            body = tpl;
            for (size_t i = 1; i < n; i++) {
                body.back() = merge_spans(body.back(), tpl[0]);
                body.insert(body.end(), tpl.begin() + 1, tpl.end());
            }
        }
        return body;
    }
    static std::vector<std::string> split_dot(const std::string& s) {
        std::vector<std::string> r; std::stringstream ss(s); std::string part;
        while (std::getline(ss, part, '.')) r.push_back(part);
        return r;
    }
    static u128 parse_value(const std::string& s) {
        u128 v = 0;
        if (s.rfind("0x", 0) == 0) { for (size_t i = 2; i < s.size(); i++) { char c = s[i]; int d = c <= '9' ? c - '0' : (c | 32) - 'a' + 10; v = v * 16 + d; } }
        else for (char c : s) v = v * 10 + (c - '0');
        return v;
    }
    void parse_op(const std::vector<std::string>& op, std::vector<UserOp>& ops, std::map<size_t, u128>& hints) {   // assembly/mod.rs:201, parsers.rs
        const std::string& name = op[0];
        int param = op.size() > 1 && name != "push" ? std::stoi(op[1]) : 1;
        if (name == "noop") ops.push_back(OP_NOOP);
        else if (name == "add") ops.push_back(OP_ADD);
        else if (name == "mul") ops.push_back(OP_MUL);
        else if (name == "swap") ops.push_back(param == 1 ? OP_SWAP : param == 2 ? OP_SWAP2 : OP_SWAP4);
        else if (name == "dup") ops.push_back(param == 1 ? OP_DUP : param == 2 ? OP_DUP2 : OP_DUP4);
        else if (name == "drop") { if (param == 4) ops.push_back(OP_DROP4); else for (int i = 0; i < param; i++) ops.push_back(OP_DROP); }
        else if (name == "push") {                                          // parsers.rs:51-62
            size_t pad = (8 - ops.size() % 8) % 8;
            ops.resize(ops.size() + pad, OP_NOOP);
            hints[ops.size()] = parse_value(op.at(1));
            ops.push_back(OP_PUSH);
        }
        else throw std::runtime_error("oracle assembler: unsupported instruction " + name);
    }
    size_t parse_block(std::vector<Block>& parent, size_t i) {            // assembly/mod.rs:54
        auto head = split_dot(tokens[i]);
        std::vector<Block> body;
        if (head[0] == "block") { i = parse_branch(body, i); Block g; g.is_span = false; g.body = body; parent.push_back(g); return i + 1; }
        if (head[0] == "repeat") {
            size_t n = std::stoul(head.at(1));
            if (n < 2) throw std::runtime_error("invalid number of iterations");
            i = parse_branch(body, i);
            Block g; g.is_span = false; g.body = repeat_block_sequence(body, n); parent.push_back(g); return i + 1;
        }
        throw std::runtime_error("oracle assembler: unsupported block " + head[0]);
    }
    size_t parse_branch(std::vector<Block>& body, size_t i) {             // assembly/mod.rs:133
        auto head = split_dot(tokens[i]);
        std::vector<UserOp> ops; std::map<size_t, u128> hints;
        if (head[0] == "begin") ops.push_back(OP_BEGIN);
        else if (head[0] != "block" && head[0] != "repeat") throw std::runtime_error("invalid block head");
        size_t first = i; i += 1;
        while (i < tokens.size()) {
            auto op = split_dot(tokens[i]);
            if (op[0] == "block" || op[0] == "repeat" || op[0] == "if" || op[0] == "while") {
                add_span(body, ops, hints, body.empty());
                i = parse_block(body, i);
            } else if (op[0] == "end") {
                if (i - first < 2) throw std::runtime_error("empty block");
                add_span(body, ops, hints, false);
                return i;
            } else { parse_op(op, ops, hints); i += 1; }
        }
        throw std::runtime_error("unmatched block");
    }
    Program compile(const std::string& src) {                             // assembly/mod.rs:19
        std::stringstream ss(src); std::string t; tokens.clear();
        while (ss >> t) tokens.push_back(t);
        if (tokens.empty() || tokens[0] != "begin" || tokens.back() != "end") throw std::runtime_error("invalid program");
        Program p; p.root.is_span = false;
        size_t i = parse_branch(p.root.body, 0);
        if (i < tokens.size() - 1) throw std::runtime_error("dangling instructions");
        p.finalize();
        return p;
    }
};

// ---- processor: decoder (processor/decoder/mod.rs) --------------------------------------------------------------
struct DecoderVM {
    size_t step = 0;
    vec op_counter;
    vec sponge_trace[4];
    u128 sponge[4] = {0, 0, 0, 0};
    vec cf_bits[3], ld_bits[5], hd_bits[2];
    std::vector<vec> ctx_stack, loop_stack;
    size_t ctx_depth, loop_depth;

    explicit DecoderVM(size_t init_len) {                                // decoder/mod.rs:39
        op_counter.assign(init_len, 0);
        for (auto& r : sponge_trace) r.assign(init_len, 0);
        for (auto& r : cf_bits) r.assign(init_len, 0);
        for (auto& r : ld_bits) r.assign(init_len, 0);
        for (auto& r : hd_bits) r.assign(init_len, 0);
        ctx_stack.push_back(vec(init_len, 0));
        ctx_depth = ctx_stack.size();
        loop_depth = 0;
    }
    size_t trace_length() const { return op_counter.size(); }
    template <class F> void for_all(F f) {
        f(op_counter);
        for (auto& r : sponge_trace) f(r);
        for (auto& r : cf_bits) f(r);
        for (auto& r : ld_bits) f(r);
        for (auto& r : hd_bits) f(r);
        for (auto& r : ctx_stack) f(r);
        for (auto& r : loop_stack) f(r);
    }
    void advance_step(bool is_user_op) {                                 // decoder/mod.rs:276
        step += 1;
        if (step >= trace_length()) { size_t nl = trace_length() * 2; for_all([&](vec& r) { r.resize(nl, 0); }); }
        op_counter[step] = is_user_op ? op_counter[step - 1] + 1 : op_counter[step - 1];
    }
    void set_op_bits(uint8_t flow, uint8_t user) {                       // decoder/mod.rs:303 (written at step - 1)
        size_t s = step - 1;
        for (int i = 0; i < 3; i++) cf_bits[i][s] = (flow >> i) & 1;
        for (int i = 0; i < 5; i++) ld_bits[i][s] = (user >> i) & 1;
        for (int i = 0; i < 2; i++) hd_bits[i][s] = (user >> (i + 5)) & 1;
    }
    void save_context() {                                                // decoder/mod.rs:327
        ctx_depth += 1;
        if (ctx_depth > MAX_CONTEXT_DEPTH) throw std::runtime_error("context stack overflow");
        if (ctx_depth > ctx_stack.size()) ctx_stack.push_back(vec(trace_length(), 0));
        for (size_t i = 1; i < ctx_stack.size(); i++) ctx_stack[i][step] = ctx_stack[i - 1][step - 1];
        ctx_stack[0][step] = sponge[0];
    }
    u128 pop_context() {                                                 // decoder/mod.rs:350
        if (ctx_depth == 0) throw std::runtime_error("context stack underflow");
        for (size_t i = 1; i < ctx_stack.size(); i++) ctx_stack[i - 1][step] = ctx_stack[i][step - 1];
        ctx_depth -= 1;
        return ctx_stack[0][step - 1];
    }
    void copy_context_stack() { for (auto& r : ctx_stack) r[step] = r[step - 1]; }   // decoder/mod.rs:366
    void copy_loop_stack() { for (auto& r : loop_stack) r[step] = r[step - 1]; }     // decoder/mod.rs:429
    void set_sponge(u128 a, u128 b, u128 c, u128 d) {                    // decoder/mod.rs:440
        sponge[0] = a; sponge[1] = b; sponge[2] = c; sponge[3] = d;
        for (int i = 0; i < 4; i++) sponge_trace[i][step] = sponge[i];
    }
    void start_block() {                                                 // decoder/mod.rs:160
        if (step % BASE_CYCLE_LENGTH != BASE_CYCLE_LENGTH - 1) throw std::runtime_error("cannot start context block: alignment");
        advance_step(false); save_context(); copy_loop_stack();
        set_op_bits(F_BEGIN, OP_NOOP);
        set_sponge(0, 0, 0, 0);
    }
    void end_block(u128 sibling_hash, bool true_branch) {                // decoder/mod.rs:172
        if (step % BASE_CYCLE_LENGTH != 0) throw std::runtime_error("cannot exit context block: alignment");
        advance_step(false);
        u128 context_hash = pop_context();
        copy_loop_stack();
        u128 block_hash = sponge[0];
        if (true_branch) { set_op_bits(F_TEND, OP_NOOP); set_sponge(context_hash, block_hash, sibling_hash, 0); }
        else { set_op_bits(F_FEND, OP_NOOP); set_sponge(context_hash, sibling_hash, block_hash, 0); }
    }
    void decode_op(UserOp op, u128 op_value) {                           // decoder/mod.rs:232
        if (op_value != 0) {
            if (op != OP_PUSH) throw std::runtime_error("op_value is non-zero for a non-PUSH operation");
            if (step % 8 != 0) throw std::runtime_error("invalid PUSH operation alignment");
        }
        advance_step(true); copy_context_stack(); copy_loop_stack();
        set_op_bits(F_HACC, op);
        sponge_apply_round(sponge, (u128)op, op_value, step - 1);        // decoder/mod.rs:448
        for (int i = 0; i < 4; i++) sponge_trace[i][step] = sponge[i];
    }
    static void fill_register(vec& r, size_t from, u128 value) { size_t to = r.size(); r.resize(from, 0); r.resize(to, value); }  // decoder/mod.rs:461
    void finalize_trace() {                                              // decoder/mod.rs:253
        u128 last = op_counter[step];
        fill_register(op_counter, step + 1, last);
        for (auto& r : cf_bits) fill_register(r, step, 1);
        for (auto& r : ld_bits) fill_register(r, step, 1);
        for (auto& r : hd_bits) fill_register(r, step, 1);
        for (auto& r : sponge_trace) { u128 v = r[step]; fill_register(r, step + 1, v); }
        for (auto& r : ctx_stack) { u128 v = r[step]; fill_register(r, step + 1, v); }
        for (auto& r : loop_stack) { u128 v = r[step]; fill_register(r, step + 1, v); }
        step = trace_length() - 1;
    }
    size_t max_ctx_stack_depth() const { return ctx_stack.size() - 1; }  // decoder/mod.rs:93
    size_t max_loop_stack_depth() const { return loop_stack.size(); }
    std::vector<vec> into_register_traces() {                            // decoder/mod.rs:120
        std::vector<vec> regs;
        regs.push_back(op_counter);
        for (auto& r : sponge_trace) regs.push_back(r);
        for (auto& r : cf_bits) regs.push_back(r);
        for (auto& r : ld_bits) regs.push_back(r);
        for (auto& r : hd_bits) regs.push_back(r);
        ctx_stack.pop_back();
        for (auto& r : ctx_stack) regs.push_back(r);
        for (auto& r : loop_stack) regs.push_back(r);
        return regs;
    }
};

// ---- processor: user stack (processor/stack/mod.rs) -----------------------------------------------------------------
struct StackVM {
    std::vector<vec> registers;
    size_t max_depth, depth, step = 0;
    StackVM(const vec& public_inputs, size_t init_len) {                 // stack/mod.rs:29
        size_t d = std::max(public_inputs.size(), MIN_STACK_DEPTH);
        for (size_t i = 0; i < d; i++) { vec r(init_len, 0); if (i < public_inputs.size()) r[0] = public_inputs[i]; registers.push_back(r); }
        max_depth = depth = public_inputs.size();
    }
    size_t trace_length() const { return registers[0].size(); }
    void advance_step() { step += 1; if (step >= trace_length()) { size_t nl = trace_length() * 2; for (auto& r : registers) r.resize(nl, 0); } }  // stack/mod.rs:653
    void copy_state(size_t start) { for (size_t i = start; i < depth; i++) registers[i][step] = registers[i][step - 1]; }                        // stack/mod.rs:608
    void shift_left(size_t start, size_t cnt) {                          // stack/mod.rs:614
        if (depth < cnt) throw std::runtime_error("stack underflow");
        for (size_t i = start; i < depth; i++) registers[i - cnt][step] = registers[i][step - 1];
        for (size_t i = depth - cnt; i < depth; i++) registers[i][step] = 0;
        depth -= cnt;
    }
    void shift_right(size_t start, size_t cnt) {                         // stack/mod.rs:631
        depth += cnt;
        if (depth > MAX_STACK_DEPTH) throw std::runtime_error("stack overflow");
        if (depth > max_depth) {
            max_depth += cnt;
            if (max_depth > registers.size()) { size_t add_n = max_depth - registers.size(); for (size_t i = 0; i < add_n; i++) registers.push_back(vec(trace_length(), 0)); }
        }
        for (size_t i = start; i < depth - cnt; i++) registers[i + cnt][step] = registers[i][step - 1];
    }
    void need(size_t d) const { if (depth < d) throw std::runtime_error("stack underflow"); }
    void execute(UserOp op, u128 push_value) {                           // stack/mod.rs:61
        advance_step();
        size_t s = step;
        switch (op) {
            case OP_BEGIN: case OP_NOOP: copy_state(0); break;
            case OP_PUSH: shift_right(0, 1); registers[0][s] = push_value; break;                       // :212
            case OP_DUP: need(1); shift_right(0, 1); registers[0][s] = registers[0][s - 1]; break;      // :249
            case OP_DUP2: need(2); shift_right(0, 2); registers[0][s] = registers[0][s - 1]; registers[1][s] = registers[1][s - 1]; break;   // :255
            case OP_DUP4: need(4); shift_right(0, 4); for (int i = 0; i < 4; i++) registers[i][s] = registers[i][s - 1]; break;
            case OP_DROP: need(1); shift_left(1, 1); break;                                             // :277
            case OP_DROP4: need(4); shift_left(4, 4); break;
            case OP_SWAP: need(2); registers[0][s] = registers[1][s - 1]; registers[1][s] = registers[0][s - 1]; copy_state(2); break;    // :287
            case OP_SWAP2: need(4); registers[0][s] = registers[2][s - 1]; registers[1][s] = registers[3][s - 1];
                           registers[2][s] = registers[0][s - 1]; registers[3][s] = registers[1][s - 1]; copy_state(4); break;
            case OP_ADD: { need(2); u128 x = registers[0][s - 1], y = registers[1][s - 1]; registers[0][s] = add(x, y); shift_left(2, 1); break; }   // :395
            case OP_MUL: { need(2); u128 x = registers[0][s - 1], y = registers[1][s - 1]; registers[0][s] = mul(x, y); shift_left(2, 1); break; }   // :403
            case OP_RESCR: { need(6); u128 st[6]; for (int i = 0; i < 6; i++) st[i] = registers[i][s - 1];                                        // :582
                             hasher_apply_round(st, s - 1); for (int i = 0; i < 6; i++) registers[i][s] = st[i]; copy_state(6); break; }
            default: throw std::runtime_error("oracle VM: unsupported operation");
        }
    }
    void finalize_trace() {                                              // stack/mod.rs:132
        size_t tl = trace_length();
        for (auto& r : registers) { r.resize(step + 1, 0); u128 v = r[step]; r.resize(tl, v); }
        step = tl - 1;
    }
    std::vector<vec> into_register_traces() { registers.resize(max_depth); return registers; }   // stack/mod.rs:144
};

// ---- processor::execute (processor/mod.rs:23-143) ---------------------------------------------------------------------
struct ExecutionTrace { std::vector<vec> registers; size_t ctx_depth, loop_depth; };

static inline void vm_close_block(DecoderVM& d, StackVM& s, u128 sibling_hash, bool true_branch) {    // processor/mod.rs:125
    d.decode_op(OP_NOOP, 0); s.execute(OP_NOOP, 0);
    d.end_block(sibling_hash, true_branch); s.execute(OP_NOOP, 0);
    for (size_t i = 0; i < HACC_NUM_ROUNDS; i++) { d.decode_op(OP_NOOP, 0); s.execute(OP_NOOP, 0); }
}
static inline void vm_execute_span(const Span& b, DecoderVM& d, StackVM& s, bool is_first) {         // processor/mod.rs:100
    if (!is_first) { d.decode_op(OP_NOOP, 0); s.execute(OP_NOOP, 0); }
    for (size_t i = 0; i < b.ops.size(); i++) {
        u128 v = 0;
        auto it = b.push_values.find(i);
        if (it != b.push_values.end()) v = it->second;
        d.decode_op(b.ops[i], v);
        s.execute(b.ops[i], v);
    }
}
static void vm_execute_blocks(const std::vector<Block>& blocks, DecoderVM& d, StackVM& s) {           // processor/mod.rs:50
    if (!blocks[0].is_span) throw std::runtime_error("first block in a sequence must be a Span block");
    vm_execute_span(blocks[0].span, d, s, true);
    for (size_t i = 1; i < blocks.size(); i++) {
        if (blocks[i].is_span) vm_execute_span(blocks[i].span, d, s, false);
        else {
            d.start_block(); s.execute(OP_NOOP, 0);                                                   // processor/mod.rs:118
            vm_execute_blocks(blocks[i].body, d, s);
            vm_close_block(d, s, 0, true);
        }
    }
}
static inline ExecutionTrace vm_execute(const Program& p, const vec& public_inputs) {                 // processor/mod.rs:23
    DecoderVM d(MIN_TRACE_LENGTH);
    StackVM s(public_inputs, MIN_TRACE_LENGTH);
    vm_execute_blocks(p.root.body, d, s);
    vm_close_block(d, s, 0, true);
    d.finalize_trace();
    s.finalize_trace();
    ExecutionTrace t;
    t.ctx_depth = d.max_ctx_stack_depth();
    t.loop_depth = d.max_loop_stack_depth();
    t.registers = d.into_register_traces();
    auto sr = s.into_register_traces();
    t.registers.insert(t.registers.end(), sr.begin(), sr.end());
    return t;
}

// the Fibonacci example program (examples/fibonacci.rs:32-47): n-th term, inputs [1, 0], 1 output
static inline std::string fibonacci_source(size_t n_terms) {
    return "begin repeat." + std::to_string(n_terms - 1) + " swap dup.2 drop add end end";
}

}  // namespace orc
