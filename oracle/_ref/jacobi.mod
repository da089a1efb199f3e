﻿!mod$ v1 sum:c713ddf36fa8a8e3
module jacobi
contains
subroutine rescal_jacobi(s1,s2,lmax,rl)
integer(4),intent(in)::s1
integer(4),intent(in)::s2
integer(4),intent(in)::lmax
real(8),intent(out)::rl(0_8:int(lmax-max(abs(s1),abs(s2)),kind=8))
end
subroutine anbncn_jacobi(a,b,lmax,an,bn,cn)
real(8),intent(in)::a
real(8),intent(in)::b
integer(4),intent(in)::lmax
real(8),intent(out)::an(0_8:int(lmax-1_4,kind=8))
real(8),intent(out)::bn(0_8:int(lmax-1_4,kind=8))
real(8),intent(out)::cn(0_8:int(lmax-1_4,kind=8))
end
end
