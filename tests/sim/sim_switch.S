/* tests/sim/sim_switch.S — TEST INFRASTRUCTURE ONLY: fiber context switch
 * (x86-64 SysV): saves callee-saved registers on the old stack, switches rsp. */
    .text
    .globl irs_sim_switch
    .type irs_sim_switch,@function
irs_sim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size irs_sim_switch, .-irs_sim_switch
    .section .note.GNU-stack,"",@progbits
